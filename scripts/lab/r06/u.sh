export PYTHONPATH=.
sed -i 's/for D in (16, 32, 64, 96, 128):/for D in (16, 32):/' scripts/lab/r06/topk_by_d.py
echo product; timeout 600 python scripts/lab/r06/topk_by_d.py 2>&1 | tail -2
echo narrow; LIBRECO_HIP_LIB=build/lab/libreco_tk_narrow.so timeout 600 python scripts/lab/r06/topk_by_d.py 2>&1 | tail -2
