cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "skinny" 2>&1 | tail -5 > gpurun_out/t_skinny.log
HCM_DEV_LIB=1 timeout 300 python tools/skinny_bench.py > gpurun_out/skinny_bench_plan.md 2>&1
timeout 900 python -m pytest tests/test_fusion_toggles_gpu.py -x -q -k "few_row" 2>&1 | tail -5 > gpurun_out/t_toggle.log
timeout 600 python bench.py --no-cpu-baseline --bf16-leg 0 --h2d-leg 0 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
timeout 600 python bench.py --config 0 --no-cpu-baseline > gpurun_out/bench_cfg0.json 2> gpurun_out/bench_cfg0.err
bash tools/chain_listing.sh r5b1 1
cat gpurun_out/t_skinny.log gpurun_out/t_toggle.log gpurun_out/skinny_bench_plan.md
