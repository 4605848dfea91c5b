"""Oracle MultimodalModel (torch CPU, fp32).  TEST INFRASTRUCTURE ONLY.

Restates multimodalIntraInterModal.py:13-416 of the reference: same ctor
arguments, same sub-module names (=> same state_dict keys and the same
construction order, so the same torch seed yields the same initial weights),
same ``forward(image, text_metadata)`` and the same 18 fusion strings.  The
fusion dispatch is written as a table of small closures rather than the
reference's if-chain; each entry cites the reference lines it restates.
"""
import torch
import torch.nn as nn

from .backbones import build_image_encoder
from .blocks import OracleGatedResidual, OracleMetaBlock, OracleTabTransformer

RGATT = "att-intramodal+residual+cross-attention-metadados"

FUSION_STRINGS = (
    "no-metadata", "no-metadata-without-mlp", "concatenation", "crossattention", "weighted",
    "gfcam", "cross-weights-after-crossattention", "metablock", "rg-att2fusefeatures", "rg-att",
    "att-intramodal", "att-intramodal+residual", "cross-attention-only",
    "residual+cross-attention-metadados", RGATT, RGATT + "+rg-att2fusefeatures",
    RGATT + "+metablock", RGATT + "+att-intramodal+residual",
)


def _classifier(in_dim, common_dim, num_classes, p):
    """fc_mlp_module (:134-146) / fc_mlp_module_after_metablock (:148-160)."""
    return nn.Sequential(
        nn.Linear(in_dim, common_dim), nn.LayerNorm(common_dim), nn.ReLU(), nn.Dropout(p),
        nn.Linear(common_dim, common_dim // 2), nn.LayerNorm(common_dim // 2), nn.ReLU(),
        nn.Dropout(p), nn.Linear(common_dim // 2, num_classes))


class OracleMultimodalModel(nn.Module):
    def __init__(self, num_classes, num_heads, device, cnn_model_name, text_model_name,
                 batch_size=32, common_dim=512, text_encoder_dim_output=512, vocab_size=91,
                 unfreeze_weights="frozen_weights", attention_mecanism="concatenation", n=2):
        super().__init__()
        self.device = device
        self.common_dim = D = common_dim
        self.num_heads = num_heads
        self.attention_mecanism = attention_mecanism
        self.n = n
        self.vocab_size = vocab_size
        self.num_classes = num_classes
        self.cnn_model_name = cnn_model_name
        self.text_model_name = text_model_name
        self.unfreeze_weights = unfreeze_weights
        self.text_encoder_dim_output = text_encoder_dim_output

        self.image_encoder, self.cnn_dim_output = build_image_encoder(
            cnn_model_name, D, unfreeze_weights)
        F_ = self.cnn_dim_output
        self.image_projector = nn.Linear(F_, D)

        if text_model_name == "one-hot-encoder":           # :57-65
            self.text_fc = nn.Sequential(
                nn.Linear(vocab_size, 256), nn.ReLU(), nn.Linear(256, 512), nn.ReLU(),
                nn.Linear(512, text_encoder_dim_output))
            self.text_encoder = None
        elif text_model_name == "tab-transformer":         # loadImageModelClassifier.py:186-200
            self.text_encoder = OracleTabTransformer([10] * 82, num_continuous=4, output_dim=85)
            self.text_encoder_dim_output = 85
            self.text_fc = None
        else:
            raise ValueError(f"Text encoder '{text_model_name}' não suportado.")
        T = self.text_encoder_dim_output
        self.text_projector = nn.Linear(T, D)

        for name in ("image_self_attention", "text_self_attention",
                     "image_cross_attention", "text_cross_attention"):   # :78-100
            setattr(self, name, nn.MultiheadAttention(embed_dim=D, num_heads=num_heads,
                                                      batch_first=False))
        self.img_gate = nn.Linear(D, D)
        self.txt_gate = nn.Linear(D, D)
        mb_in_common = attention_mecanism == RGATT + "+metablock"          # :112-115
        self.meta_block = OracleMetaBlock(
            V_dim=D if mb_in_common else F_,
            U_dim=D if (mb_in_common or attention_mecanism == "metablock-se") else T)
        self.image_residual = OracleGatedResidual(D)
        self.text_residual = OracleGatedResidual(D)
        self.fc_fusion = _classifier(D * (1 if attention_mecanism == "no-metadata" else n),
                                     D, num_classes, 0.5)
        self.fc_visual_only = nn.Linear(F_, num_classes)
        self.fc_fusion_proj_feat2output = nn.Linear(D, num_classes)
        self.fc_mlp_module_after_metablock_fusion_module = _classifier(F_, D, num_classes, 0.3)

    # ------------------------------------------------------------------ forward
    def forward(self, image, text_metadata):
        feat = self.image_encoder(image.to(self.device))
        if feat.dim() == 4:
            feat = feat.mean(dim=(-2, -1))
        if self.text_model_name == "one-hot-encoder":
            tfeat = self.text_fc(text_metadata.to(self.device))
        else:
            # The reference wiring for tab-transformer cannot run; the
            # build's contract: first n_cat columns are category ids, the rest continuous.
            ncat = self.text_encoder.num_categorical
            m = text_metadata.to(self.device)
            tfeat = self.text_encoder(m[:, :ncat].long(), m[:, ncat:].float())
        Pi = self.image_projector(feat).unsqueeze(0)      # (1, B, D)
        Pt = self.text_projector(tfeat).unsqueeze(0)

        isa, tsa = self.image_self_attention, self.text_self_attention
        ica, tca = self.image_cross_attention, self.text_cross_attention
        ires, tres = self.image_residual, self.text_residual
        att = lambda mod, q, kv: mod(q, kv, kv)[0]
        cat = lambda a, b: torch.cat([a.squeeze(0), b.squeeze(0)], dim=1)
        fcf = self.fc_fusion
        f2o = self.fc_fusion_proj_feat2output
        sig = torch.sigmoid

        def self_att():                                   # :193-194
            return att(isa, Pi, Pi), att(tsa, Pt, Pt)

        def cross_after_self():                           # :193-200
            Ai, At = self_att()
            return att(ica, Ai, At), att(tca, At, Ai)

        def rgatt_core():                                 # :322-336
            Ai, At = self_att()
            Ri, Rt = ires(Pi, Ai, Ai), tres(Pt, At, At)
            return att(ica, Ri, Rt), att(tca, Rt, Ri)

        def m_gate(swap):                                 # :225-235
            Ci, Ct = cross_after_self()
            ai, at_ = sig(self.img_gate(Ci)), sig(self.txt_gate(Ct))
            return fcf(cat(at_ * Ci, ai * Ct)) if swap else fcf(cat(ai * Ci, at_ * Ct))

        def m_res_cross():                                # :301-318
            Ri, Rt = ires(Pi, Pi, Pi), tres(Pt, Pt, Pt)
            return fcf(cat(att(ica, Ri, Rt), att(tca, Rt, Ri)))

        def m_rgatt_rg2():                                # :343-361
            Xi, Xt = rgatt_core()
            return f2o(ires(Xt, Xi, Xi).squeeze(0))

        def m_rgatt_mb():                                 # :364-386
            Xi, Xt = rgatt_core()
            return f2o(self.meta_block(Xi.squeeze(0), Xt.squeeze(0)))

        def m_rgatt_again():                              # :388-412
            Xi, Xt = rgatt_core()
            A2i, A2t = att(isa, Xi, Xi), att(tsa, Xt, Xt)
            return fcf(cat(ires(Xi, A2i, A2i), tres(Xt, A2t, A2t)))

        table = {
            "no-metadata": lambda: fcf(Pi.squeeze(0)),                                   # :205
            "no-metadata-without-mlp": lambda: self.fc_visual_only(feat),               # :208
            "concatenation": lambda: fcf(cat(Pi, Pt)),                                  # :211
            "crossattention": lambda: fcf(cat(*cross_after_self())),                    # :215
            "weighted": lambda: fcf(cat(sig(self.img_gate(Pi)) * Pi,
                                        sig(self.txt_gate(Pt)) * Pt)),                  # :219
            "gfcam": lambda: m_gate(False),
            "cross-weights-after-crossattention": lambda: m_gate(True),
            "metablock": lambda: self.fc_mlp_module_after_metablock_fusion_module(
                self.meta_block(feat, tfeat)),                                           # :237
            "rg-att2fusefeatures": lambda: f2o(ires(Pt, Pi, Pi).squeeze(0)),            # :247
            "rg-att": lambda: fcf(cat(ires(Pi, Pt, Pt), tres(Pt, Pi, Pi))),             # :253
            "att-intramodal": lambda: fcf(cat(*self_att())),                            # :265
            "att-intramodal+residual": lambda: (lambda Ai, At: fcf(cat(
                ires(Pi, Ai, Ai), tres(Pt, At, At))))(*self_att()),                     # :273
            "cross-attention-only": lambda: fcf(cat(att(ica, Pi, Pt), att(tca, Pt, Pi))),  # :285
            "residual+cross-attention-metadados": m_res_cross,
            RGATT: lambda: fcf(cat(*rgatt_core())),
            RGATT + "+rg-att2fusefeatures": m_rgatt_rg2,
            RGATT + "+metablock": m_rgatt_mb,
            RGATT + "+att-intramodal+residual": m_rgatt_again,
        }
        fn = table.get(self.attention_mecanism)
        if fn is None:
            raise ValueError(f"Attention mechanism '{self.attention_mecanism}' not implemented.")
        return fn()
