mkdir -p gpurun_out/r6r
python tools/mfma_calib.py loop > gpurun_out/r6r/mfma_loop.txt 2>&1
python tools/host_procs.py 8 8 2.0 > gpurun_out/r6r/host_procs.json 2> gpurun_out/r6r/host_procs.err
(time python bench.py --steps 20 --warmup 5 > gpurun_out/r6r/bench.json 2> gpurun_out/r6r/bench.err) 2> gpurun_out/r6r/bench_time.txt
tail -n 6 gpurun_out/r6r/mfma_loop.txt; cat gpurun_out/r6r/host_procs.json; cat gpurun_out/r6r/bench_time.txt
python -c "import json;d=json.load(open('gpurun_out/r6r/bench.json'));print(d['value'],d['ms_per_step']);print(d.get('host_8proc'))"
