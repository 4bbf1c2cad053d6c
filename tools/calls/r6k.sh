mkdir -p gpurun_out/r6k
for i in 1 2 3; do python -m pytest tests/test_fullsize_gpu.py -x -q -k "config4_full_size_properties" 2>&1 | grep -E "passed|failed"; done > gpurun_out/r6k/cfg4.txt
python -m pytest tests/test_ops_gpu.py -x -q -k "stem" > gpurun_out/r6k/t_ops.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r6k/bench.json 2> gpurun_out/r6k/bench.err
python -m pytest tests -q -m gpu -x --deselect tests/test_ops_gpu.py > gpurun_out/r6k/t_all.txt 2>&1
cat gpurun_out/r6k/cfg4.txt; tail -n 3 gpurun_out/r6k/t_ops.txt; tail -n 12 gpurun_out/r6k/t_all.txt
python -c "import json;d=json.load(open('gpurun_out/r6k/bench.json'));print(d['value'],d['ms_per_step'],d['bf16_mode']['value'])"
