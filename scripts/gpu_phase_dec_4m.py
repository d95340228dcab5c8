"""Decoder phase breakdown of 4 MiB blocks (configs[2]'s shape: 64 consecutive corpus blocks as one block), diagnostic
build (LZ4MI_LIB=lz4_amd/liblz4mi_prof.so).  usage: [NB] [first corpus block]"""
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd import _native as N
from lz4_amd.device import DeviceCodec
from tools.corpus import corpus
from scripts.phase_names import names
import os, subprocess, tempfile
from lz4_amd import Lz4FrameBlockSize, Lz4FrameOptions
from lz4_amd.sharding import frame_block_table
from scripts.bench_configs import _cli
codec = DeviceCodec(0)
L = N.load()
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 64
data0 = corpus(64 * NB)
with tempfile.TemporaryDirectory(dir="/tmp") as td:   # the frame the lz4 CLI writes for these bytes (real LZ4 matches)
    pin, pout = os.path.join(td, "in.bin"), os.path.join(td, "out.lz4")
    data0.tofile(pin)
    r = subprocess.run([_cli(), "-f", "-q", "-B7", "-BX", "--content-size", pin, pout], capture_output=True)
    assert r.returncode == 0, r.stderr
    cli_frame = np.fromfile(pout, dtype=np.uint8)
fbytes = cli_frame.tobytes()
info, offs, end = frame_block_table(fbytes)
dopts = Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k4MB, blockChecksum=info["block_checksum"])
first = int(offs[0])
d_records = torch.from_numpy(cli_frame[first:info["body_end"]]).cuda()
rel = torch.from_numpy(offs - first).cuda()
slots = torch.empty((len(offs), 4 << 20), dtype=torch.uint8, device="cuda")
codec.shard_decode(d_records, rel, dopts, out=slots); torch.cuda.synchronize()
buf = (C.c_ulonglong * 64)()
L.lz4mi_debug_dec_phases(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); codec.shard_decode(d_records, rel, dopts, out=slots); e1.record(); torch.cuda.synchronize()
assert np.array_equal(slots.reshape(-1)[:data0.size].cpu().numpy(), data0)
L.lz4mi_debug_dec_phases(buf, 1)
v = list(buf)
print(f'{NB} blocks of 4 MiB from the CLI (ratio {len(fbytes) / data0.size:.3f}): {e0.elapsed_time(e1):.3f} ms', flush=True)
for role, title in ((0, 'parsing wavefront'), (1, 'copying wavefront')):
    print(f'  {title}')
    for i, nm in enumerate(names):
        x = v[32 * role + i]
        if x:
            print(f'   {nm:34s} {x / NB:10.0f} per block' if 'count' in nm else f'   {nm:34s} {x / NB / 1000:10.1f} kcycles per block')
