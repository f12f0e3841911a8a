for args in "--batch 4" "--batch 8" "--size 2048 --steps 5" "--deform-mode deformable" "--compute f32_split2" "--compute f32_split2 --batch 4" "--compute bf16 --batch 8"; do
  python bench.py $args --cpu-baseline off --also "" --profile-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$args', '->', d['value'], 'img/s', d['ms_per_step'], 'ms')"
done
