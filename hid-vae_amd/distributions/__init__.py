"""Mirror of the reference's distributions/ package (distributions/gumbel.py)."""
