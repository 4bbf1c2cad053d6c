mkdir -p gpurun_out/r6h
cd robo-vln_amd/csrc
cp ../libhcm_dev.so /tmp/libhcm_dev_fast.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhcm_dev.so build_dev_on/*.o -ldl
cd ../..
for k in NONE HCM_NO_BNECK_FUSE HCM_NO_BNECK256 HCM_NO_BNECK_NEXT HCM_NO_GN_ONLOAD HCM_NO_DEPTH_L3 HCM_NO_DEPTH_BLK HCM_NO_SKINNY HCM_NO_VLA_FUSE HCM_NO_BNECK_DSFOLD HCM_NO_STEM_FUSE HCM_NO_GN_RES2 HCM_NO_GN_POOL HCM_SERIAL; do
  r=$(env HCM_DEV_LIB=1 $k=1 python -m pytest tests/test_fullsize_gpu.py -x -q -k "config4_full_size_properties" 2>&1 | grep -E "passed|failed")
  echo "$k=1: $r" >> gpurun_out/r6h/knobs.txt
done
cp /tmp/libhcm_dev_fast.so robo-vln_amd/libhcm_dev.so
cat gpurun_out/r6h/knobs.txt
