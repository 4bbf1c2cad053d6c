mkdir -p gpurun_out/r6t
python -m pytest tests/test_fullsize_gpu.py tests/test_fusion_toggles_gpu.py -q > gpurun_out/r6t/t_sub.txt 2>&1
python bench.py > gpurun_out/r6t/bench_default.json 2> gpurun_out/r6t/bench_default.err
bash tools/chain_listing.sh r6t 64 > /dev/null 2>&1
tail -n 3 gpurun_out/r6t/t_sub.txt; ls gpurun_out/r6t
python -c "import json;d=json.load(open('gpurun_out/r6t/bench_default.json'));print(d['value'],d['ms_per_step'],d['sustained']['value'],d['bf16_mode']['value'],d['single_env_latency']['chain_graphs'],{k:v.get('value') for k,v in d['configs'].items()})"
