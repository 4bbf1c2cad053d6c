mkdir -p gpurun_out/r6l
python bench.py --steps 20 --warmup 5 > gpurun_out/r6l/bench.json 2> gpurun_out/r6l/bench.err
python -m pytest tests -q -m gpu --deselect tests/test_ops_gpu.py > gpurun_out/r6l/t_all.txt 2>&1
python -m pytest tests/test_ops_gpu.py -q > gpurun_out/r6l/t_ops.txt 2>&1
tail -n 25 gpurun_out/r6l/t_all.txt | cut -c1-300; tail -n 5 gpurun_out/r6l/t_ops.txt
python -c "import json;d=json.load(open('gpurun_out/r6l/bench.json'));print(d['value'],d['ms_per_step'],d['bf16_mode']['value']);print(d.get('gather_world1'));print(d.get('configs'))"
