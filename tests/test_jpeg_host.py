"""Host side of the device JPEG writer: marker segments and tables equal Pillow's (no GPU needed)."""
import io
import numpy as np
import pytest
from PIL import Image


@pytest.mark.parametrize("quality", [1, 30, 75, 95, 100])
def test_header_equals_pillow(quality):
    from imagetransformations_amd import jpeg
    a = np.zeros((37, 53, 3), np.uint8)
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, "JPEG", quality=quality)
    hdr = jpeg.header(53, 37, quality)
    assert len(hdr) == 623 and buf.getvalue()[:623] == hdr


def test_tables_match_the_restatement():
    from imagetransformations_amd import jpeg
    from oracle import jpeg_oracle as J
    t = jpeg.tables(75)
    lum, chr_ = J.quant_tables(75)
    assert list(t.quant[0]) == list(lum) and list(t.quant[1]) == list(chr_)
    ac = J.huff_codes(J.AC_LUM_BITS, J.AC_LUM_VALS)
    for sym, (code, length) in ac.items():
        assert (t.ac_code[0][sym], t.ac_len[0][sym]) == (code, length)
    dc = J.huff_codes(J.DC_CHR_BITS, J.DC_VALS)
    for sym, (code, length) in dc.items():
        assert (t.dc_code[1][sym], t.dc_len[1][sym]) == (code, length)
    with pytest.raises(ValueError):
        jpeg.header(70000, 10)
