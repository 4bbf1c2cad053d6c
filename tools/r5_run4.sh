cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for nj in 1 2 3 5; do HCM_DEV_LIB=1 HCM_SKINNY_NJ=$nj timeout 300 python tools/skinny_bench.py > gpurun_out/skinny_bench_nj$nj.md 2>&1; done
HCM_DEV_LIB=1 timeout 600 python tools/step_marks.py 1 > gpurun_out/marks_b1.txt 2>&1
for b in 1 2 4; do timeout 600 python tools/act_host_profile.py $b 2>&1 | grep "^B="; done > gpurun_out/host_b1.txt
paste -d'|' <(cut -d'|' -f2,3,5 gpurun_out/skinny_bench_nj1.md) <(cut -d'|' -f5 gpurun_out/skinny_bench_nj2.md) <(cut -d'|' -f5 gpurun_out/skinny_bench_nj3.md) <(cut -d'|' -f5 gpurun_out/skinny_bench_nj5.md)
cat gpurun_out/marks_b1.txt | head -50; cat gpurun_out/host_b1.txt
