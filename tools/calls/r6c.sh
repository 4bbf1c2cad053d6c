mkdir -p gpurun_out/r6c
python -m pytest tests/test_ops_gpu.py -x -q -k "stem" > gpurun_out/r6c/t_ops.txt 2>&1
python -m pytest tests/test_fusion_toggles_gpu.py -x -q -k "one_launch_stem or simplecnn or three_conv" > gpurun_out/r6c/t_tog.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r6c/bench.json 2> gpurun_out/r6c/bench.err
python -m pytest tests -q -m gpu -x --deselect tests/test_ops_gpu.py > gpurun_out/r6c/t_all.txt 2>&1
tail -n 3 gpurun_out/r6c/t_ops.txt gpurun_out/r6c/t_tog.txt
tail -n 15 gpurun_out/r6c/t_all.txt
python -c "import json;d=json.load(open('gpurun_out/r6c/bench.json'));print(d['value'],d['ms_per_step'],d['bf16_mode']['value'])"
