"""Host logic of imagetransformations_amd.staging that needs no GPU: PIL images mapped onto RGBX blocks."""
import gc

import numpy as np
from PIL import Image


def test_zero_copy_pil_images_share_the_block_and_release_their_budget():
    from imagetransformations_amd import staging
    rng = np.random.default_rng(0)
    block = rng.integers(0, 256, (3, 37, 53, 4), dtype=np.uint8)
    before = staging._zc_live
    assert staging.zero_copy_reserve(block.nbytes)
    imgs = [staging.image_from_rgbx(block[j]) for j in range(3)]
    assert staging._zc_live == before + block.nbytes
    for j, im in enumerate(imgs):
        assert im.mode == "RGB" and im.size == (53, 37)
        assert np.array_equal(np.asarray(im), block[j, :, :, :3])
        assert np.array_equal(np.asarray(im.copy()), block[j, :, :, :3])
        assert im.tobytes() == block[j, :, :, :3].tobytes()
    # a change goes into a private copy (the mapped image is read-only), the block is untouched
    keep = block.copy()
    imgs[0].putpixel((0, 0), (1, 2, 3))
    assert np.array_equal(block, keep) and np.asarray(imgs[0])[0, 0].tolist() == [1, 2, 3]
    # ordinary Pillow operations accept it
    assert imgs[1].rotate(10).size == (53, 37) and imgs[1].resize((8, 8)).mode == "RGB"
    del imgs, im
    gc.collect()
    assert staging._zc_live == before
    # beyond the budget the drivers fall back to Image.fromarray
    assert not staging.zero_copy_reserve(staging.ZERO_COPY_BUDGET + 1)
