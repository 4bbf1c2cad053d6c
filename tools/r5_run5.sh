cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/t_full.log
HCM_DEV_LIB=1 timeout 300 python tools/skinny_bench.py > gpurun_out/skinny_bench_final.md 2>&1
timeout 600 python bench.py --no-cpu-baseline --bf16-leg 0 --h2d-leg 0 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
timeout 600 python bench.py --config 0 --no-cpu-baseline > gpurun_out/bench_cfg0.json 2> gpurun_out/bench_cfg0.err
cat gpurun_out/t_full.log gpurun_out/skinny_bench_final.md
