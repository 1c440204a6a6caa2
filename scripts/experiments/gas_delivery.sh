#!/bin/bash
# "gas" output: every gas's lines call delivers its block piece by piece ("each") against only the
# last one ("last"), interleaved, 1 to 16 levels (scripts/perf_api_levels.py).
for round in 1 2; do
for mode in last each; do
  PYLBL_AMD_GAS_DELIVERY=$mode python scripts/perf_api_levels.py 1 4 8 16 2>/dev/null | sed "s/^/$mode round $round /"
done
done
