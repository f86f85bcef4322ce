for n in 256 512 768 1024 1280 2560; do
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-lossless --no-single-clip --no-shard --no-e2e --clips-per-gpu $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($n, d['roofline']['kernel_ms'])"
done
