"""Seeded synthetic inputs of the hot path's shapes (bench.py, tests/, tools/, oracle/gen_golden.py): data generators only, no arithmetic of
the path and nothing the product imports."""
