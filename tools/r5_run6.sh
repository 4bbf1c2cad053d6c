cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo "== $*"; env HCM_DEV_LIB=1 "$@" timeout 300 python tools/act_host_profile.py 1 2>&1 | grep "chain_graphs=True"; }
{
run A=1
run HCM_NO_DEPTH_BLK=1
run HCM_NO_DEPTH_L3=1
run HCM_NO_DEPTH_BLK=1 HCM_NO_DEPTH_L3=1
run HCM_NO_BNECK_FUSE=1
run HCM_NO_BNECK_NEXT=1
run HCM_NO_BNECK256=1
run HCM_NO_VLA_FUSE=1
run HCM_NO_GN_ONLOAD=1
run HCM_NO_SKINNY=1
run HCM_SKINNY_MAX_ROWS=320
} > gpurun_out/b1_toggles.txt 2>&1
cat gpurun_out/b1_toggles.txt
