# bisect: which translation unit, compiled with -ffp-contract=on, makes test_config4_full_size_properties disagree with itself
mkdir -p gpurun_out/r6g
cd robo-vln_amd/csrc
cp ../libhcm_dev.so /tmp/libhcm_dev_fast.so
for f in igemm gemm256 vla_fused simplecnn elementwise attention depth_blk bert_block skinny stem; do
  objs=""
  for g in igemm gemm256 vla_fused simplecnn elementwise attention depth_blk bert_block skinny stem weights forward api comm; do
    if [ "$g" = "$f" ]; then objs="$objs build_dev_on/$g.o"; else objs="$objs build_dev/$g.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhcm_dev.so $objs -ldl
  r=$(cd ../.. && HCM_DEV_LIB=1 python -m pytest tests/test_fullsize_gpu.py -x -q -k "config4_full_size_properties" 2>&1 | grep -E "passed|failed")
  echo "$f on, rest fast: $r" >> ../../gpurun_out/r6g/bisect.txt
done
cp /tmp/libhcm_dev_fast.so ../libhcm_dev.so
cat ../../gpurun_out/r6g/bisect.txt
