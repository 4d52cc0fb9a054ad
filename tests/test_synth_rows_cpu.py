"""synth.device_rows_chunks draws the rows of synth.device_rows again, a chunk at a time: bench.py frees C4's 30 GB of rows once
the store holds them and its full-size parity check streams them from the generator a second time.  The two must agree bit
for bit — on the CPU generator here, on the device's in the GPU tier."""
import pytest
import torch

from meilisearch_amd import synth


def _check(dev):
    n, d, chunk = 2500, 24, 1000
    whole = synth.device_rows(n, d, dev, seed=77, chunk=chunk)
    seen = 0
    for c0, c1, rows in synth.device_rows_chunks(n, d, dev, seed=77, chunk=chunk):
        assert c0 == seen and c1 == min(n, c0 + chunk) and rows.shape == (c1 - c0, d)
        assert torch.equal(rows.view(torch.int32), whole[c0:c1].view(torch.int32))
        seen = c1
    assert seen == n


def test_chunks_are_the_rows_again_on_the_cpu_generator():
    _check(torch.device("cpu"))


@pytest.mark.gpu
def test_chunks_are_the_rows_again_on_the_device_generator():
    _check(torch.device("cuda:0"))
