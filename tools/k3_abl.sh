# tools/k3_ubench variants over the dominant shapes of configs[3] (gpurun -- 'bash tools/k3_abl.sh "<variant suffixes>"')
mkdir -p gpurun_out/r6_k3abl
V=${1:-"_f0 _f1 _f2 _f3"}
while read sh; do
for n in $V; do [ $n = _ ] && n=""; echo -n "[$n] "; timeout 60 tools/k3_ubench$n $sh | tr '\n' ' ' | sed 's/TFLOP.s algorithmic//; s/output digest//; s/K3_ABL=0 //'; echo; done
done 2>&1 <<SHAPES | tee gpurun_out/r6_k3abl/abl_$(date +%H%M%S).log
128 8 8 512 512 2 w1
128 64 64 64 128 2 w1
128 32 32 128 256 2 w1
128 16 16 256 512 2 w1
128 16 16 256 256 2 w2
128 32 32 128 128 2 w2
SHAPES
