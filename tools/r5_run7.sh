cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -k "attention or bert_attn or gelu or skinny" 2>&1 | tail -5 > gpurun_out/t_att.log
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_bf16_margin_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/t_par.log
timeout 900 python bench.py --no-cpu-baseline --h2d-leg 0 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cat gpurun_out/t_att.log gpurun_out/t_par.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['bf16_mode']['value'], d['bf16_mode']['ms_per_step'])
PY
