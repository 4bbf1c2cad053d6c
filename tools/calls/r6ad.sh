#!/bin/bash
# round 6, call 31: the whole GPU suite + smoke with the register-weights cross-modal layer as the default
mkdir -p gpurun_out/r6ad
timeout 2700 python -m pytest tests -q -m gpu -x > gpurun_out/r6ad/gpu_suite.log 2>&1; tail -4 gpurun_out/r6ad/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
