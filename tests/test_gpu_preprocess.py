"""-m gpu: the input side of the path -- A.Resize and the OneHotEncoder + StandardScaler metadata encoding
on the GPU, through the C ABI.  Resize: bit-exact against oracle/preprocess.py (numpy restatement of cv2's published 8-bit
INTER_LINEAR algorithm; cv2 itself is absent -> parity unpinned) and within one grey level of float bilinear.  Metadata:
against sklearn's own OneHotEncoder / StandardScaler, the reference's dependency (skinLesionDatasets.py:155-180)."""
import numpy as np
import pytest
import torch

from gpu_util import DEV
from mmskin.preprocess import MetadataEncoder, resize_u8
from oracle.preprocess import resize_bilinear_float, resize_linear_u8

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("src,dst", [((300, 400), (224, 224)), ((112, 96), (224, 224)), ((448, 448), (224, 224)),
                                     ((225, 223), (224, 224)), ((64, 64), (7, 9)), ((1, 5), (4, 4)), ((224, 224), (224, 224))])
def test_resize_u8_matches_cv2_restatement(src, dst):
    g = torch.Generator().manual_seed(src[0] * 1000 + src[1])
    img = torch.randint(0, 256, (3, src[0], src[1], 3), generator=g, dtype=torch.uint8)
    got = resize_u8(img.to(DEV), dst).cpu().numpy()
    want = resize_linear_u8(img.numpy(), dst[0], dst[1])
    assert got.shape == want.shape == (3, dst[0], dst[1], 3)
    assert np.array_equal(got, want), int(np.abs(got.astype(int) - want.astype(int)).max())
    assert np.abs(got.astype(float) - resize_bilinear_float(img.numpy(), dst[0], dst[1])).max() < 1.0


def test_resize_then_model_equals_host_transform():
    """Val/test transform on the device end to end: raw uint8 batch of another size -> Resize -> Normalize -> ToTensor ->
    model, against the same steps done on the host feeding the fp32 NCHW tensor (skinLesionDatasets.py:116-120)."""
    import os
    from helpers import SMALL
    from models import multimodalIntraInterModal as M
    from oracle.detinit import det_init_, det_inputs
    os.environ["MMSKIN_BACKBONE_DTYPE"] = "fp32"
    model = det_init_(M.MultimodalModel(**dict(SMALL, cnn_model_name="resnet-18", attention_mecanism="concatenation", device=DEV))).to(DEV).eval()
    g = torch.Generator().manual_seed(5)
    raw = torch.randint(0, 256, (3, 150, 200, 3), generator=g, dtype=torch.uint8)
    meta = det_inputs(3, 32, 20, 6)[1].to(DEV)
    model.image_encoder.resize_to = (96, 96)
    with torch.no_grad():
        a = model(raw.to(DEV), meta).cpu()
        host = torch.from_numpy(resize_linear_u8(raw.numpy(), 96, 96)).float() / 255.0
        host = ((host - torch.tensor([0.485, 0.456, 0.406])) / torch.tensor([0.229, 0.224, 0.225])).permute(0, 3, 1, 2).contiguous()
        b = model(host.to(DEV), meta).cpu()
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), (a - b).abs().max()


def test_metadata_encode_matches_sklearn():
    from sklearn.preprocessing import OneHotEncoder, StandardScaler
    rng = np.random.default_rng(0)
    n = 257                                              # PAD-UFES-20-like: string categoricals + 3 numerics with gaps
    cats = np.stack([rng.choice(["True", "False", "EMPTY"], n), rng.choice(["ARM", "FACE", "BACK", "CHEST", "EMPTY"], n),
                     rng.choice(["FEMALE", "MALE", "EMPTY"], n), rng.choice(["POMERANIA", "GERMANY", "BRAZIL", "EMPTY", "ITALY"], n)], axis=1)
    num = np.stack([rng.integers(6, 95, n).astype(float), rng.uniform(1, 40, n), rng.uniform(1, 30, n)], axis=1)
    num[rng.random((n, 3)) < 0.15] = np.nan             # pd.to_numeric(errors="coerce") leaves NaN -> fillna(-1)
    num_filled = np.where(np.isnan(num), -1.0, num)
    ohe = OneHotEncoder(sparse_output=False, handle_unknown="ignore").fit(cats[:200])
    sc = StandardScaler().fit(num_filled[:200])
    want = np.hstack([ohe.transform(cats), sc.transform(num_filled)])        # skinLesionDatasets.py:160-183
    for enc in (MetadataEncoder().fit(cats[:200], num[:200]), MetadataEncoder.from_sklearn(ohe, sc)):
        assert enc.width == want.shape[1]
        assert all(list(a) == list(b) for a, b in zip(enc.categories_, ohe.categories_))
        got = enc.transform(enc.codes(cats).to(DEV), torch.from_numpy(num).float().to(DEV)).cpu().numpy()
        assert got.shape == want.shape
        assert np.array_equal(got[:, :enc.onehot_width], want[:, :enc.onehot_width])      # one-hot block: exact
        assert np.allclose(got[:, enc.onehot_width:], want[:, enc.onehot_width:], rtol=1e-5, atol=1e-5)
    # a category unseen at fit time encodes as all zeros in its block (handle_unknown='ignore')
    odd = cats[:2].copy(); odd[0, 1] = "SCALP"
    got = enc.transform(enc.codes(odd).to(DEV), torch.from_numpy(num[:2]).float().to(DEV)).cpu().numpy()
    assert np.array_equal(got[:, :enc.onehot_width], ohe.transform(odd))
    # a constant numeric column keeps scale 1 like sklearn
    const = MetadataEncoder().fit(cats[:50], np.ones((50, 2)))
    assert np.allclose(const.scale_, StandardScaler().fit(np.ones((50, 2))).scale_)
