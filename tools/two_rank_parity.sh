for i in 1 2 3; do
  SDV_FORCE_DEVICE=0 python bench.py --gpus 2 --arch tiny --steps 1 --warmup 1 --batch-size 4 --no-cpu-baseline --no-walk-pass --no-kernel-pass 2>&1 | grep -o '"parity_check": {[^}]*}' | head -1
done
