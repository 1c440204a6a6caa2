# Does the pedestal leg of the default bench depend on what ran before it?  (one run had 4.76 ms
# against 4.21-4.23 everywhere else)
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'step', round(d['ms_per_step'],3), 'ped', round(d['pedestal_option']['ms_per_step'],3), 'banded', round(d['banded_table_option']['ms_per_step'],3) if 'banded_table_option' in d else '')"; }
for rep in 1 2; do
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --extras pedestal 2>/dev/null | show "s20 pedestal only rep$rep"
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --extras sustained,pedestal 2>/dev/null | show "s20 sustained+pedestal rep$rep"
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --extras overlap,pedestal 2>/dev/null | show "s20 overlap+pedestal rep$rep"
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --extras sustained,overlap,pedestal,banded 2>/dev/null | show "s20 sus+overlap+pedestal+banded rep$rep"
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --extras sustained,overlap,pedestal,banded 2>/dev/null | show "s10 sus+overlap+pedestal+banded rep$rep"
done
