import sys, faulthandler
faulthandler.dump_traceback_later(25, exit=True)
sys.path.insert(0, '.')
def P(*a): print(*a, flush=True)
import torch
import lz4_amd
case = int(sys.argv[1])
cases = [
    (b"", 0),
    (bytes([0x50, 0x48, 0x65, 0x6c, 0x6c, 0x6f]), 5),
    (bytes([0x40, 0x61, 0x62, 0x63, 0x64, 0x04, 0x00]), 8),
    (bytes([0x13, 0x41, 0x01, 0x00]), 8),
    (bytes([0xF0, 0x05]) + bytes(range(20)), 20),
    (bytes([0x00]), 4),
]
src, size = cases[case]
P('case', case)
try:
    out = lz4_amd.lz4Decompress(src, decompressedSize=size)
    P('out', out)
except Exception as e:
    P('exc', type(e).__name__, e)
