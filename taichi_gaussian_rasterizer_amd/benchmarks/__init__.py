"""Per-operator benchmarks with the reference's entry points and flags (reference benchmarks/bench_*.py,
run by its tests/test_benchmarks.py): `bench_projection`, `bench_sh`, `bench_tilemapper`, `bench_rasterizer`,
each with `parse_args(args=None)` and `bench_<op>(args)`.  Timing is by HIP events around `iters` calls; every
`bench_<op>` returns {phase name: milliseconds per call}.  The frame-level benchmark is `bench.py` at the repo root."""
