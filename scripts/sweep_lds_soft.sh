for soft in 0 50000 76000; do
  echo "== XRFTHIP_LDS_SOFT=$soft"
  if [ $soft = 0 ]; then unset XRFTHIP_LDS_SOFT; else export XRFTHIP_LDS_SOFT=$soft; fi
  timeout 200 python scripts/prof_generic.py 2>&1 | grep "GFFT\|main\]\|us/slab" | head -40
done
