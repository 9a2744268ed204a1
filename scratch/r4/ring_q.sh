#!/bin/bash
for q in 0 8 10 11 12 15 16 17 20 22 24 32; do
  echo "== Q=$q"; env HIDVAE_RING_Q=$q python scratch/r4/lbwd_time.py 2>/dev/null | grep -E "1024 x  (691|768|460|512|345) x"
done
