export TMPDIR=/tmp
mkdir -p gpurun_out/phase
cp gms_amd/lib/libgmsx.so /tmp/keep.so
for ph in 7 1 2 4; do
  if [ $ph = 7 ]; then cp /tmp/keep.so gms_amd/lib/libgmsx.so; else cp gms_amd/lib/variants/libgmsx_p$ph.so gms_amd/lib/libgmsx.so; fi
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/phase/p$ph -o pmc -- python3 tools/tc_phase_probe.py 26 > gpurun_out/phase/out$ph.json 2> gpurun_out/phase/err$ph.txt
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/phase/f$ph -o pmc -- python3 tools/tc_phase_probe.py 26 >> gpurun_out/phase/out$ph.json 2>> gpurun_out/phase/err$ph.txt
  cat gpurun_out/phase/out$ph.json
done
cp /tmp/keep.so gms_amd/lib/libgmsx.so
