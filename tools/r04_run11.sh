#!/bin/bash
O=gpurun_out/r04k; mkdir -p $O
TAG=both python tools/probes/phase_probe.py 2>&1 | grep -v amdgpu | tee $O/phase_both.txt
TAG=am_only OSP_TAPE_VOC=0 python tools/probes/phase_probe.py 2>&1 | grep -v amdgpu | tee $O/phase_am.txt
TAG=voc_only OSP_TAPE_AM=0 python tools/probes/phase_probe.py 2>&1 | grep -v amdgpu | tee $O/phase_voc.txt
