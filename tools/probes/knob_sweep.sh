#!/bin/bash
# schedule knobs A/B on one box: 3 x 120 pipelined steps each (tools/step_profile.py) -> min / median / max ms per step
run() { tag=$1; shift; xs=""; for i in 1 2 3; do x=$(env "$@" STEPS=120 OSP_PIPELINE_STEPS=1 python tools/step_profile.py 2>/dev/null | grep -o "[0-9.]* ms per step" | cut -d" " -f1); xs="$xs $x"; done; echo "$tag: $(echo $xs | tr ' ' '\n' | sort -n | tr '\n' ' ')"; }
run default X=1
run steps_ahead_1 OSP_MAX_STEPS_AHEAD=1
run steps_ahead_3 OSP_MAX_STEPS_AHEAD=3
run steps_ahead_4 OSP_MAX_STEPS_AHEAD=4
run wgrad_flush_2 OSP_WGRAD_FLUSH=2
run wgrad_flush_4 OSP_WGRAD_FLUSH=4
run wgrad_inline OSP_WGRAD_STREAM=0
run d_after_g OSP_D_AFTER_G=1
run disc_streams_6 OSP_DISC_MAX_STREAMS=6
run tape_am OSP_TAPE_SEGMENTS=1 OSP_TAPE_VOC=0
run tape_both OSP_TAPE_SEGMENTS=1
run no_tapes OSP_TAPES=0
run no_fused_mlp OSP_FUSED_MLP=0
run default_again X=1
