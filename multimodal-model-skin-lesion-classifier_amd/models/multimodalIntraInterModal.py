"""MultimodalModel on the MI355X HIP path -- drop-in for the reference's
src/scripts/benchmark/models/multimodalIntraInterModal.py:13-416.

Same constructor arguments (positional order included), same ``forward(image, text_metadata)``,
same 18 fusion strings, same sub-module names (=> state_dict keys, and the same construction order, so
the same torch seed yields the same initial head weights as the reference) and the same exceptions.
PyTorch supplies nn.Parameter storage, autograd bookkeeping and device memory; every arithmetic step
of forward/backward is a HIP kernel behind the C ABI of include/mmskin.h.

Differences that are deliberate:
  * the four seq_len=1 attention modules are evaluated lazily, only when the chosen fusion string
    consumes them (the reference always runs them and throws the results away, :193-197); outputs
    and gradients are unchanged (unused modules get grad None either way);
  * ``tab-transformer`` metadata is runnable (the reference wiring raises):
    ``text_metadata`` is a float tensor whose first 82 columns are category ids and last 4 continuous.
"""
import os
import sys

import torch
import torch.nn as nn

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
for _p in (_PKG, _HERE):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from gatedResidualBlock import GatedAlteredResidualBlock, StackedGatedResidualBlock  # noqa: E402,F401
from loadImageModelClassifier import loadModels  # noqa: E402
from metablock import MetaBlock  # noqa: E402
from mmskin import ops  # noqa: E402
from mmskin.nn import FusedAway, HipDropout, HipLayerNorm, HipLinear, HipMultiheadAttention  # noqa: E402

_RG = "att-intramodal+residual+cross-attention-metadados"


class MultimodalModel(nn.Module):
    def __init__(self, num_classes, num_heads, device, cnn_model_name, text_model_name, batch_size=32,
                 common_dim=512, text_encoder_dim_output=512, vocab_size=91,
                 unfreeze_weights="frozen_weights", attention_mecanism="concatenation", n=2):
        super().__init__()
        self.device = device
        self.common_dim = common_dim
        self.num_heads = num_heads
        self.attention_mecanism = attention_mecanism
        self.n = n
        self.vocab_size = vocab_size
        self.num_classes = num_classes
        self.cnn_model_name = cnn_model_name
        self.text_model_name = text_model_name
        self.unfreeze_weights = unfreeze_weights
        self.text_encoder_dim_output = text_encoder_dim_output

        self.image_encoder, self.cnn_dim_output = loadModels.loadModelImageEncoder(
            cnn_model_name=cnn_model_name, common_dim=common_dim, backbone_train_mode=unfreeze_weights)
        self.image_projector = HipLinear(self.cnn_dim_output, common_dim)

        if text_model_name == "one-hot-encoder":
            self.text_fc = nn.Sequential(
                HipLinear(vocab_size, 256, fuse_relu=True), FusedAway("ReLU"),
                HipLinear(256, 512, fuse_relu=True), FusedAway("ReLU"),
                HipLinear(512, text_encoder_dim_output))
            self.text_encoder = None
        else:
            self.text_encoder, self.text_encoder_dim_output, _ = loadModels.loadTextModelEncoder(
                text_model_encoder=text_model_name, train_mode=unfreeze_weights)
            self.text_fc = None
        self.text_projector = HipLinear(self.text_encoder_dim_output, common_dim)

        mha = lambda: HipMultiheadAttention(embed_dim=common_dim, num_heads=num_heads, batch_first=False)
        self.image_self_attention = mha()
        self.text_self_attention = mha()
        self.image_cross_attention = mha()
        self.text_cross_attention = mha()

        self.img_gate = HipLinear(common_dim, common_dim)
        self.txt_gate = HipLinear(common_dim, common_dim)

        in_common = attention_mecanism == _RG + "+metablock"
        self.meta_block = MetaBlock(
            V_dim=common_dim if in_common else self.cnn_dim_output,
            U_dim=common_dim if (in_common or attention_mecanism == "metablock-se")
            else self.text_encoder_dim_output)
        self.image_residual = GatedAlteredResidualBlock(dim=common_dim)
        self.text_residual = GatedAlteredResidualBlock(dim=common_dim)

        self.fc_fusion = self.fc_mlp_module(n=1 if attention_mecanism == "no-metadata" else n)
        self.fc_visual_only = HipLinear(self.cnn_dim_output, num_classes)
        self.fc_fusion_proj_feat2output = HipLinear(common_dim, num_classes)
        self.fc_mlp_module_after_metablock_fusion_module = self.fc_mlp_module_after_metablock()

    # classifier heads: Linear-LN-ReLU-Drop-Linear-LN-ReLU-Drop-Linear (indices 0,1,4,5,8 hold params)
    def _head(self, in_dim, p):
        D = self.common_dim
        return nn.Sequential(
            HipLinear(in_dim, D), HipLayerNorm(D, fuse_relu=True), FusedAway("ReLU"), HipDropout(p),
            HipLinear(D, D // 2), HipLayerNorm(D // 2, fuse_relu=True), FusedAway("ReLU"), HipDropout(p),
            HipLinear(D // 2, self.num_classes))

    def fc_mlp_module(self, n=1):
        return self._head(self.common_dim * n, 0.5)

    def fc_mlp_module_after_metablock(self):
        return self._head(self.cnn_dim_output, 0.3)

    # ------------------------------------------------------------------------------ forward
    def _metadata_features(self, text_metadata):
        if self.text_model_name == "one-hot-encoder":
            return self.text_fc(text_metadata.to(self.device))
        if self.text_model_name == "tab-transformer":
            m = text_metadata.to(self.device)
            ncat = self.text_encoder.num_categorical
            return self.text_encoder(m[:, :ncat].long(), m[:, ncat:].float())
        input_ids = text_metadata["input_ids"].squeeze(1).to(self.device)
        attention_mask = text_metadata["attention_mask"].squeeze(1).to(self.device)
        return self.text_encoder(input_ids=input_ids, attention_mask=attention_mask).last_hidden_state[:, 0, :]

    def forward(self, image, text_metadata):
        mech = self.attention_mecanism
        img_feat = self.image_encoder(image.to(self.device))
        if img_feat.dim() == 4:
            img_feat = img_feat.mean(dim=(-2, -1))
        needs_meta = mech not in ("no-metadata", "no-metadata-without-mlp")
        if mech == "no-metadata-without-mlp":
            return self.fc_visual_only(img_feat)
        txt_feat = self._metadata_features(text_metadata) if needs_meta else None
        if mech == "metablock":
            return self.fc_mlp_module_after_metablock_fusion_module(self.meta_block(img_feat, txt_feat))

        P_i = self.image_projector(img_feat).unsqueeze(0)            # (1, B, D)
        if mech == "no-metadata":
            return self.fc_fusion(P_i.squeeze(0))
        P_t = self.text_projector(txt_feat).unsqueeze(0)

        isa, tsa = self.image_self_attention, self.text_self_attention
        ica, tca = self.image_cross_attention, self.text_cross_attention
        ires, tres = self.image_residual, self.text_residual
        att = lambda mod, q, kv: mod(q, kv, kv)[0]
        fuse = lambda a, b: self.fc_fusion(ops.concat2(a.squeeze(0), b.squeeze(0)))
        f2o = self.fc_fusion_proj_feat2output

        if mech == "concatenation":
            return fuse(P_i, P_t)
        if mech == "weighted":
            return fuse(ops.sigmoid_gate(self.img_gate(P_i), P_i), ops.sigmoid_gate(self.txt_gate(P_t), P_t))
        if mech == "rg-att2fusefeatures":
            return f2o(ires(P_t, P_i, P_i).squeeze(0))
        if mech == "rg-att":
            return fuse(ires(P_i, P_t, P_t), tres(P_t, P_i, P_i))
        if mech == "cross-attention-only":
            return fuse(att(ica, P_i, P_t), att(tca, P_t, P_i))
        if mech == "residual+cross-attention-metadados":
            R_i, R_t = ires(P_i, P_i, P_i), tres(P_t, P_t, P_t)
            return fuse(att(ica, R_i, R_t), att(tca, R_t, R_i))

        # everything below starts from the intra-modal self-attention of both streams (:193-194)
        A_i, A_t = att(isa, P_i, P_i), att(tsa, P_t, P_t)
        if mech == "att-intramodal":
            return fuse(A_i, A_t)
        if mech == "att-intramodal+residual":
            return fuse(ires(P_i, A_i, A_i), tres(P_t, A_t, A_t))
        if mech in ("crossattention", "gfcam", "cross-weights-after-crossattention"):
            C_i, C_t = att(ica, A_i, A_t), att(tca, A_t, A_i)        # :196-200
            if mech == "crossattention":
                return fuse(C_i, C_t)
            z_i, z_t = self.img_gate(C_i), self.txt_gate(C_t)
            if mech == "gfcam":
                return fuse(ops.sigmoid_gate(z_i, C_i), ops.sigmoid_gate(z_t, C_t))
            return fuse(ops.sigmoid_gate(z_t, C_i), ops.sigmoid_gate(z_i, C_t))
        if mech.startswith(_RG):
            R_i, R_t = ires(P_i, A_i, A_i), tres(P_t, A_t, A_t)
            X_i, X_t = att(ica, R_i, R_t), att(tca, R_t, R_i)
            tail = mech[len(_RG):]
            if tail == "":
                return fuse(X_i, X_t)
            if tail == "+rg-att2fusefeatures":
                return f2o(ires(X_t, X_i, X_i).squeeze(0))
            if tail == "+metablock":
                return f2o(self.meta_block(X_i.squeeze(0), X_t.squeeze(0)))
            if tail == "+att-intramodal+residual":
                A2_i, A2_t = att(isa, X_i, X_i), att(tsa, X_t, X_t)
                return fuse(ires(X_i, A2_i, A2_i), tres(X_t, A2_t, A2_t))
        raise ValueError(f"Attention mechanism '{mech}' not implemented.")
