"""Regenerates the lz4-CLI fixture (needs the `lz4` binary; v1.9.3 in the build image).

Input: the first 8 blocks of the synthetic corpus (tools/corpus.c); output: the frame
written by `lz4 -B4 -BX` (64 KiB independent blocks with block checksums)."""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tools.corpus import corpus  # noqa: E402

cli = "/opt/conda/bin/lz4"
with tempfile.NamedTemporaryFile(delete=False) as f:
    f.write(corpus(8).tobytes())
out = subprocess.run([cli, "-c", "-f", "-B4", "-BX", f.name], capture_output=True, check=True).stdout
open(os.path.join(HERE, "lz4cli_v193_corpus8_B4_BX.lz4"), "wb").write(out)
os.unlink(f.name)
print(len(out))
