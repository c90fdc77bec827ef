set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/final/tests.log 2>&1; echo "tests rc=$?" ; tail -2 gpurun_out/final/tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/final/default.json 2> gpurun_out/final/default.err; echo "default rc=$?"
python bench.py --steps 20 --warmup 5 --gemm bf16 --no-cpu-baseline > gpurun_out/final/bf16.json 2> gpurun_out/final/bf16.err; echo "bf16 rc=$?"
python bench.py --batch 32 --points 2048 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/final/b32.json 2>/dev/null
python bench.py --batch 32 --points 2048 --steps 50 --warmup 10 --no-cpu-baseline --graph > gpurun_out/final/b32g.json 2>/dev/null
python bench.py --batch 512 --points 2048 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/final/b512.json 2>/dev/null
python bench.py --batch 512 --points 2048 --steps 10 --warmup 3 --no-cpu-baseline --gemm bf16 > gpurun_out/final/b512_bf16.json 2>/dev/null
python bench.py --batch 4096 --points 2048 --steps 5 --warmup 2 --no-cpu-baseline --gemm bf16 > gpurun_out/final/b4096n2048_bf16.json 2>/dev/null
for f in default bf16 b32 b32g b512 b512_bf16 b4096n2048_bf16; do echo $f; cut -c90-260 gpurun_out/final/$f.json; done
