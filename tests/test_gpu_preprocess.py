"""GPU parity: §8(f)-1 prepare_image (normalise + TF bilinear resize + pad), bit-exact vs the oracle."""
import numpy as np
import pytest
import torch

import oracle as o

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h,w,target", [(480, 640, 640), (427, 640, 640), (1000, 333, 640), (37, 53, 128), (640, 640, 640),
                                        (2000, 3000, 1024)])
def test_prepare_image_bit_exact(cuda, params, h, w, target):
    from retinanet.dataloader.preprocessing_pipeline import PreprocessingPipeline
    rng = np.random.default_rng(h * 7 + w)
    img = rng.integers(0, 256, (h, w, 3)).astype(np.float32)
    pp = params.dataloader_params
    pipe = PreprocessingPipeline([target, target], pp)
    out = pipe.normalize_and_resize_with_pad(torch.from_numpy(img).to(cuda))
    torch.cuda.synchronize()
    want, scale = o.prepare_image(img, target, target, pp.preprocessing.mean, pp.preprocessing.stddev,
                                  pp.preprocessing.pixel_scale)
    np.testing.assert_array_equal(out["resize_scale"].numpy(), scale)
    np.testing.assert_array_equal(out["image"].cpu().numpy().view(np.uint32), want.view(np.uint32))
    # identity resize keeps the normalised pixels exactly; padding is zero
    if h == w == target:
        ref = (img / np.float32(255.0) - np.float32(pp.preprocessing.mean)) / np.float32(pp.preprocessing.stddev)
        np.testing.assert_array_equal(out["image"].cpu().numpy(), ref.astype(np.float32))
