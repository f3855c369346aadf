#!/bin/bash
# Which mix reproduces the shared-GPU FFT deviation?  Pairs of tools/component_race_probe.py on one GPU.
N=${1:-150}
pair() {   # $1 label, $2 args of process A, $3 args of process B
  echo "== $1"
  ( timeout 600 python tools/component_race_probe.py f32 $N $2 2>&1 | grep "iterations;" | sed "s/^/A: /" ) > /tmp/_ma.txt &
  ( timeout 600 python tools/component_race_probe.py f32 $N $3 2>&1 | grep "iterations;" | sed "s/^/B: /" ) > /tmp/_mb.txt &
  wait; cat /tmp/_ma.txt /tmp/_mb.txt
}
pair "both processes: STFT kernels + torch.stft only (no conv stacks in either)" stft-only stft-only
pair "A: full mix (STFT + conv stacks), B: STFT only" "" stft-only
pair "both: full mix" "" ""
pair "both: full mix, every component synchronised before the next (shallow queue)" sync-each sync-each
