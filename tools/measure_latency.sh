set -o pipefail
export PYTHONUNBUFFERED=1
O=gpurun_out/r02b; mkdir -p $O
python bench.py > $O/bench_chacha20.json 2> $O/bench_chacha20.err && echo "bench chacha20 ok" &&
for b in 1 16 32 64; do python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_chacha20_b$b.json 2> $O/bench_chacha20_b$b.err && echo "bench b$b ok"; done &&
python bench.py --workload aes128 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_aes128.json 2> $O/bench_aes128.err && echo "aes ok" &&
python bench.py --workload aes128 --batch 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_aes128_b1.json 2> $O/bench_aes128_b1.err && echo "aes b1 ok" &&
python bench.py --workload mixed --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_mixed.json 2> $O/bench_mixed.err && echo "mixed ok"
python tools/prof_few_scaling.py > $O/few_scaling.txt 2>&1
cat $O/*.json | cut -c1-400
