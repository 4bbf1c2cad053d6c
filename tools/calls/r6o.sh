mkdir -p gpurun_out/r6o
python tools/stem_fused_bench.py 64 > gpurun_out/r6o/stem_bench.txt 2>&1
python tools/stem_fused_bench.py 1 >> gpurun_out/r6o/stem_bench.txt 2>&1
python -m pytest tests/test_ops_gpu.py -x -q -k "stem" 2>&1 | tail -n 2 > gpurun_out/r6o/t_ops.txt
python -m pytest tests/test_fusion_toggles_gpu.py -x -q -k "one_launch_stem" 2>&1 | tail -n 2 >> gpurun_out/r6o/t_ops.txt
python tools/stem_determinism.py 128 40 > gpurun_out/r6o/det.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r6o/bench.json 2> gpurun_out/r6o/bench.err
cat gpurun_out/r6o/stem_bench.txt gpurun_out/r6o/t_ops.txt gpurun_out/r6o/det.txt
python -c "import json;d=json.load(open('gpurun_out/r6o/bench.json'));print(d['value'],d['ms_per_step'],d['bf16_mode']['value'],d['single_env_latency']['chain_graphs'])"
