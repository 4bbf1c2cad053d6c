mkdir -p gpurun_out/r3i
export HCM_DEV_LIB=1
for i in 1 2; do
 for v in none low high; do
  if [ $v = none ]; then unset HCM_AUX_PRIO; else export HCM_AUX_PRIO=$v; fi
  python bench.py --no-cpu-baseline --sustain 0 --steps 100 --no-kernel-probe 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
  HCM_GRAPH=0 python bench.py --no-cpu-baseline --sustain 0 --steps 100 --no-kernel-probe 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v nograph', d['value'], d['ms_per_step'])"
 done
done
unset HCM_AUX_PRIO; unset HCM_DEV_LIB
cp robo-vln_amd/libhcm.so /tmp/new.so
for i in 1 2; do
  cp robo-vln_amd/libhcm_prev.so robo-vln_amd/libhcm.so; python bench.py --no-cpu-baseline --sustain 0 --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prev', d['value'], d['ms_per_step'], d['roofline']['frac'])"
  cp /tmp/new.so robo-vln_amd/libhcm.so; python bench.py --no-cpu-baseline --sustain 0 --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
