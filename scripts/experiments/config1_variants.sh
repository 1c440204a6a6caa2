for extra in "" "--engine-option aligned_tiles=1 --points-per-lane 8" "--engine-option aligned_tiles=1 --points-per-lane 4" "--engine-option aligned_tiles=1 --points-per-lane 2" "--engine-option item_floor=1024" "--engine-option item_floor=4096" "--engine-option lanes=4" ; do
  for rep in 1 2; do
  line=$(python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline --config 1 $extra 2>/dev/null | tail -1)
  python -c "import json,sys; d=json.loads(sys.argv[1]); print('%-60s ms/step %.4f value %.4g' % (sys.argv[2], d['ms_per_step'], d['value']))" "$line" "$extra"
  done
done
