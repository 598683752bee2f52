mkdir -p gpurun_out
run() {
  out=$(env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-extras --no-cpu --blocks 262144 --dist $DIST 2>>gpurun_out/sweep8.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); e=d['extras']; k=list(e.keys())[0]; print('enc', e[k]['encode_fast_GBps'], 'dec', d['value'], d['verified'])")
  echo "dist=$DIST $* -> $out" | tee -a gpurun_out/sweep8.txt
}
for DIST in 2 3; do
  run LZ4HIP_ENCODER=sm LZ4HIP_ENCODER_WAVES_PER_CU=8
  run LZ4HIP_ENCODER=sm LZ4HIP_ENCODER_WAVES_PER_CU=16
  run LZ4HIP_ENCODER=sm LZ4HIP_ENCODER_WAVES_PER_CU=32
done
DIST=1; run LZ4HIP_ENCODER=sm LZ4HIP_ENCODER_WAVES_PER_CU=16
DIST=0; run LZ4HIP_ENCODER=sm LZ4HIP_ENCODER_WAVES_PER_CU=16
