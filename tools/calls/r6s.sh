mkdir -p gpurun_out/r6s
python -m pytest tests -q -m gpu > gpurun_out/r6s/t_all.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6s/smoke.txt 2>&1
tail -n 8 gpurun_out/r6s/t_all.txt | cut -c1-300; tail -n 2 gpurun_out/r6s/smoke.txt
