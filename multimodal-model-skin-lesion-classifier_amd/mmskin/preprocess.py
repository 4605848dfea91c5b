"""Input side of the path on the GPU: what the reference's Dataset does on the host before
`model(image, metadata)` -- `A.Resize` of the val/test transform (skinLesionDatasets.py:116-120) and the
OneHotEncoder + StandardScaler metadata encoding (skinLesionDatasets.py:133-183).

    enc = MetadataEncoder().fit(categorical_rows, numeric_rows)      # or MetadataEncoder.from_sklearn(ohe, scaler)
    codes = enc.codes(categorical_rows)                              # host: strings -> int32 category indices
    meta = enc.transform(codes.cuda(), numeric.cuda())               # GPU: [B, onehot_width + n_num] fp32
    images = resize_u8(raw_u8_nhwc.cuda(), (224, 224))               # GPU: A.Resize; Normalize + ToTensor happen in the stem

Strings never reach the GPU: mapping a value to its index among the column's fitted categories stays on the host (one
dict lookup per cell); everything numeric runs through the C ABI.  No CPU fallback.
"""
import numpy as np
import torch

from . import ops
from ._lib import call, ptr, stream


def resize_u8(images, size):
    """uint8 NHWC [N, Hs, Ws, 3] -> uint8 NHWC [N, size[0], size[1], 3] with cv2.resize(INTER_LINEAR) semantics."""
    ops._need_gpu(images, "resize_u8")
    if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[-1] != 3:
        raise ValueError(f"resize_u8 expects uint8 NHWC [N, H, W, 3], got {images.dtype} {tuple(images.shape)}")
    images = images.contiguous()
    n, hs, ws, _ = images.shape
    h, w = int(size[0]), int(size[1])
    if (hs, ws) == (h, w):
        return images
    out = torch.empty((n, h, w, 3), dtype=torch.uint8, device=images.device)
    call("mmskin_resize_u8", ptr(images), n, hs, ws, ptr(out), h, w, stream())
    return out


class MetadataEncoder:
    """sklearn OneHotEncoder(sparse_output=False, handle_unknown='ignore') + StandardScaler, as the reference fits them
    (skinLesionDatasets.py:155-180): categories are the sorted unique strings of each column, the scaler uses the
    population standard deviation (ddof 0) with scale 1 for constant columns; missing numerics are filled with -1 BEFORE
    fitting and transforming (:148)."""

    nan_fill = -1.0

    def __init__(self):
        self.categories_ = None
        self.mean_ = None
        self.scale_ = None
        self._dev = {}

    # ---- fitting (host, once per dataset)
    def fit(self, categorical_rows, numeric_rows):
        cats = np.asarray(categorical_rows, dtype=object)
        cats = cats.reshape(len(cats), -1) if cats.size else np.empty((len(numeric_rows), 0), dtype=object)
        self.categories_ = [np.array(sorted({str(v) for v in cats[:, j]}), dtype=object) for j in range(cats.shape[1])]
        num = np.asarray(numeric_rows, dtype=np.float64).reshape(len(cats), -1)
        num = np.where(np.isnan(num), self.nan_fill, num)
        self.mean_ = num.mean(axis=0)
        var = num.var(axis=0)
        scale = np.sqrt(var)
        scale[scale < 10 * np.finfo(np.float64).eps * np.maximum(np.abs(self.mean_), 1.0)] = 1.0    # sklearn: _handle_zeros_in_scale
        self.scale_ = scale
        self._dev = {}
        return self

    @classmethod
    def from_sklearn(cls, ohe, scaler):
        """Adopt fitted sklearn objects (the reference pickles them under ./data/preprocess_data)."""
        self = cls()
        self.categories_ = [np.asarray(c, dtype=object) for c in ohe.categories_]
        self.mean_ = np.asarray(scaler.mean_, dtype=np.float64)
        self.scale_ = np.asarray(scaler.scale_, dtype=np.float64)
        return self

    @property
    def onehot_width(self):
        return int(sum(len(c) for c in self.categories_))

    @property
    def width(self):
        return self.onehot_width + len(self.mean_)

    # ---- per batch
    def codes(self, categorical_rows):
        """Host: [B, n_cat] int32 indices into the fitted categories; -1 for a value unseen at fit time."""
        rows = np.asarray(categorical_rows, dtype=object)
        rows = rows.reshape(len(rows), -1)
        lut = [{str(v): i for i, v in enumerate(c)} for c in self.categories_]
        out = np.full(rows.shape, -1, dtype=np.int32)
        for j, table in enumerate(lut):
            out[:, j] = [table.get(str(v), -1) for v in rows[:, j]]
        return torch.from_numpy(out)

    def _tables(self, device):
        t = self._dev.get(str(device))
        if t is None:
            off = np.concatenate([[0], np.cumsum([len(c) for c in self.categories_])]).astype(np.int32)
            t = (torch.from_numpy(off).to(device), torch.tensor(self.mean_, dtype=torch.float32, device=device),
                 torch.tensor(self.scale_, dtype=torch.float32, device=device))
            self._dev[str(device)] = t
        return t

    def transform(self, codes, numeric):
        """GPU: codes int32 [B, n_cat], numeric fp32 [B, n_num] (NaN = missing) -> fp32 [B, width]."""
        ops._need_gpu(codes, "metadata_encode")
        codes = codes.to(torch.int32).contiguous()
        numeric = numeric.to(device=codes.device, dtype=torch.float32).contiguous()
        b, n_cat = codes.shape
        n_num = numeric.shape[1]
        if n_cat != len(self.categories_) or n_num != len(self.mean_):
            raise ValueError(f"metadata_encode: fitted for {len(self.categories_)} categorical + {len(self.mean_)} numeric "
                             f"columns, got {n_cat} + {n_num}")
        off, mean, scale = self._tables(codes.device)
        out = torch.empty((b, self.width), dtype=torch.float32, device=codes.device)
        call("mmskin_metadata_encode", ptr(codes), n_cat, ptr(off), self.onehot_width, ptr(numeric), n_num, ptr(mean),
             ptr(scale), float(self.nan_fill), ptr(out), b, stream())
        return out
