"""TEST INFRASTRUCTURE ONLY.  CPU restatements of the HiD-VAE tokenizer hot path used to check
the HIP product.  Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg."""
