python -m pytest tests/test_ops_gpu.py -q -k "bottleneck or bneck or fold or down" 2>&1 | tail -2
cp robo-vln_amd/libhcm.so /tmp/new.so
for i in 1 2; do
  cp robo-vln_amd/libhcm_prev.so robo-vln_amd/libhcm.so; python bench.py --no-cpu-baseline --sustain 0 --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prev', d['value'], d['ms_per_step'], d['roofline']['frac'])"
  cp /tmp/new.so robo-vln_amd/libhcm.so; python bench.py --no-cpu-baseline --sustain 0 --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
