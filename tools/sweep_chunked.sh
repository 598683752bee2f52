mkdir -p gpurun_out
run() {
  out=$(env "$@" timeout 200 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu --blocks 524288 --dist $DIST 2>>gpurun_out/sweep4.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['mean_kernel_ms'], d['verified'])")
  echo "dist=$DIST $* -> $out" | tee -a gpurun_out/sweep4.txt
}
for DIST in 2 3; do
  run LZ4HIP_DECODER=chunked LZ4HIP_STAGE_BYTES=128
  run LZ4HIP_DECODER=chunked LZ4HIP_STAGE_BYTES=256
done
