# GPU session r02ce: device memory per input byte (contexts of their own, device-resident buffers not counted)
python tools/diag/mem.py
exit 0
