"""Parts of bench.py (the entry point at the repository root): kernel timings, the other configurations, the CPU baseline, the
launcher."""
