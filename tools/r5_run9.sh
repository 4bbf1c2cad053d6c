cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
b() { env HCM_DEV_LIB=1 "$@" timeout 600 python bench.py --no-cpu-baseline --bf16-leg 0 --h2d-leg 0 --latency-leg 0 --no-kernel-probe --sustain 0 --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
{
for i in 1 2; do
echo "default"; b A=1
echo "LN_FOLD=2"; b HCM_LN_FOLD=2
echo "LN_FOLD=1"; b HCM_LN_FOLD=1
done
} > gpurun_out/lnfold_ab.txt 2>&1
cat gpurun_out/lnfold_ab.txt
