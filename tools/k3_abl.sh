mkdir -p gpurun_out/r6_k3abl
for sh in "96 8 8 512 512" "96 64 64 128 64" "96 16 16 512 256"; do
for n in "" _1 _2 _4 _8 _3 _7 _14 _13 _11; do timeout 60 tools/k3_ubench$n $sh; done
done 2>&1 | tee gpurun_out/r6_k3abl/abl.log
timeout 900 python -m pytest tests/test_gpu_dp_stub_collective.py tests/test_gpu_fanogan.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
