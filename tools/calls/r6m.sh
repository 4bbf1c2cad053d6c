mkdir -p gpurun_out/r6m
python -m pytest tests/test_trained_like_weights_gpu.py -q -s > gpurun_out/r6m/t_trained.txt 2>&1
bash tools/ab.sh r6m/prio_bert "HCM_STREAM_PRIO=0,0,-1,0" 3 > gpurun_out/r6m/ab_prio_bert.txt 2>&1
bash tools/ab.sh r6m/prio_bert_depthlow "HCM_STREAM_PRIO=0,1,-1,0" 2 > gpurun_out/r6m/ab_prio2.txt 2>&1
grep -E "trained-like|passed|failed|Error" gpurun_out/r6m/t_trained.txt | cut -c1-600
cat gpurun_out/r6m/ab_prio_bert.txt gpurun_out/r6m/ab_prio2.txt
