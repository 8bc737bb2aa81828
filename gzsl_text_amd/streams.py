"""Second-stream concurrency helper for independent branches of the train step."""
import os

import torch


class Beside:
    """Runs the replicated label encoder on a second HIP stream beside the note encoder (independent until the
    attention).  The label side is a chain of small launches (spmm, N x 3H GEMMs, gate kernels) that leave most CUs
    idle, and the note encoder's big GEMMs leave part of their last round empty: together they fill the chip.

    Ordering rule that makes the caching allocator safe without record_stream: the side stream waits for the caller's
    stream at entry (so memory it reuses is no longer read by earlier work) and the caller's stream waits for the side
    stream at `join()` (so everything produced there is visible, and later frees are ordered after its readers)."""

    _streams = {}
    serial = False  # set by HipOps(deterministic=True)

    def __init__(self, ref_tensor):
        # (deterministic mode: both branches add into the word-table gradient -- one stream fixes their order)
        self.on = (ref_tensor.is_cuda and os.environ.get('GZSL_NO_SIDE_STREAM') != '1'
                   and os.environ.get('GZSL_DETERMINISTIC') != '1' and not Beside.serial)
        if self.on:
            dev = ref_tensor.device
            self.main = torch.cuda.current_stream(dev)
            self.side = Beside._streams.get(dev)
            if self.side is None:
                self.side = Beside._streams[dev] = torch.cuda.Stream(dev)

    def __enter__(self):
        if self.on:
            self.side.wait_stream(self.main)
            self._ctx = torch.cuda.stream(self.side)
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self._ctx.__exit__(*exc)
        return False

    def join(self):
        if self.on:
            self.main.wait_stream(self.side)
