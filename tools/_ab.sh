for mb in 512 1024 2048 4096; do
  python bench.py --mode infer --micro-batch $mb --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('infer mb $mb', d['value'], d['ms_per_step'])"
done
