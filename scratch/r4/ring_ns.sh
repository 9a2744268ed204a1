#!/bin/bash
for e in "HIDVAE_RING_NS=4" "HIDVAE_RING_NS=3" "HIDVAE_RING_NS=3 HIDVAE_RING_MINQ=6" "HIDVAE_RING_NS=4 HIDVAE_RING_MINQ=12"; do
  echo "== $e"; env $e python scratch/r4/lbwd_time.py 2>/dev/null | grep " x "
done
