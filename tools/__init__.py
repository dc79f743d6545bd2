"""Dev tools and bench.py's helper modules (bench_common, bench_extra, bench_multirank): a regular package, so that no `tools` elsewhere on sys.path can shadow it."""
