"""CPU: the NumPy Philox4x32-10 (oracle/philox_oracle.py) against Random123's known-answer vectors."""
import numpy as np

from oracle import philox_oracle as PO


def test_philox4x32_10_known_answers():
    for ctr, key, want in PO.KAT:
        got = PO.philox4x32_10(np.array(ctr, np.uint32), np.array(key, np.uint32))
        assert [int(v) for v in got] == list(want), (ctr, key)


def test_stream_layout():
    s = PO.stream(16, seed=0)
    assert [int(v) for v in s[:4]] == list(PO.KAT[0][2])            # block 0 = counter (0,0,0,0), key (0,0)
    assert np.array_equal(PO.stream(8, 0x0123456789abcdef, offset=8), PO.stream(16, 0x0123456789abcdef)[8:])
