"""Synthetic mixed-entropy corpus (tests + bench input; not part of the product)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libcorpus.so")
_lib = None


def build():
    src = os.path.join(_HERE, "corpus.c")
    if not os.path.exists(_PATH) or os.path.getmtime(_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libcorpus.so"], stdout=subprocess.DEVNULL)
    return _PATH


def _load():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_PATH)
        L.corpus_fill.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_size_t, C.c_void_p]
        L.corpus_fill.restype = None
        L.corpus_fill(0, 0, 0, 0, None)  # initialise the vocabulary before going multi-threaded
        _lib = L
    return _lib


def corpus(n_blocks: int, block: int = 65536, first: int = 0, seed: int = 0, threads: int = 8) -> np.ndarray:
    """uint8 array of n_blocks * block bytes; block b has class (first + b) % 8."""
    L = _load()
    out = np.empty(n_blocks * block, dtype=np.uint8)
    base = out.ctypes.data
    chunk = max(1, min(256, (n_blocks + threads - 1) // threads))

    def work(lo):
        cnt = min(chunk, n_blocks - lo)
        L.corpus_fill(first + lo, cnt, seed, block, C.c_void_p(base + lo * block))

    if n_blocks <= chunk or threads <= 1:
        for lo in range(0, n_blocks, chunk):
            work(lo)
    else:
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(work, range(0, n_blocks, chunk)))
    return out


def corpus_block(index: int, block: int = 65536, seed: int = 0) -> np.ndarray:
    return corpus(1, block, first=index, seed=seed, threads=1)
