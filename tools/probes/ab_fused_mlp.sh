for i in 1 2; do for v in 0 1; do echo -n "FUSED_MLP=$v: "; OSP_FUSED_MLP=$v STEPS=40 OSP_PIPELINE_STEPS=1 python tools/step_profile.py 2>&1 | tail -1; done; done
