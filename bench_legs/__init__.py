"""The legs of bench.py, one module each, every function taking an explicit context (bench_legs.common.Ctx) instead of
closing over bench.main()'s locals.  bench.py owns the argument parser, the launcher, the headline's timed region and the
compact result line; everything a leg measures goes into the detail file (bench_detail.json)."""
