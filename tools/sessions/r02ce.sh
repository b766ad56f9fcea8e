# GPU session r02ce: device memory per input byte (contexts of their own, device-resident buffers not counted)
python tools/device_memory.py
exit 0
