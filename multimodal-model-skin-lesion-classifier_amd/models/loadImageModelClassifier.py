"""Encoder factory on the HIP path -- drop-in for models/loadImageModelClassifier.py:9-203.

Same static-method API, argument meaning, return values and error strings as the reference's
``loadModels``.  Image backbones with a gfx950 plan: ``custom-cnn``, ``resnet-18``, ``resnet-50``,
``densenet169``, ``vgg16``, ``mobilenet-v2``, ``efficientnet-b0/b7``.
Weights are randomly initialised (torchvision layout and init); pretrained checkpoints are loaded by
the caller with ``load_state_dict`` -- there is no network access from this package.
"""
import os
import sys

import torch.nn as nn

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (_PKG, os.path.dirname(os.path.abspath(__file__))):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from mmskin.backbone import RESNET_DEPTHS, HipCustomCNN, HipDenseNet, HipEfficientNet, HipMobileNetV2, HipResNet, HipVGG16  # noqa: E402
from tab_transformer import TabTransformer  # noqa: E402

# Backbones the reference accepts (loadImageModelClassifier.py:65-118) that have no HIP plan yet.
_KNOWN_WITHOUT_PLAN = ()


class loadModels:

    @staticmethod
    def set_backbone_train_mode(model, mode="frozen_weights", last_n_layers=1):
        """reference :15-35 -- freeze everything, then re-enable per `mode`."""
        params = list(model.parameters())
        for p in params:
            p.requires_grad = False
        if mode == "frozen_weights":
            return
        if mode == "unfrozen_weights":
            for p in params:
                p.requires_grad = True
        elif mode == "last_layer_unfrozen_weights":
            for p in params[-last_n_layers * 2:]:
                p.requires_grad = True
        else:
            raise ValueError(f"Invalid backbone_train_mode: {mode}")

    @staticmethod
    def loadModelImageEncoder(cnn_model_name: str, common_dim: int, backbone_train_mode: str = "frozen",
                              device: str = "cpu"):
        """reference :41-157 -> (nn.Module, cnn_dim_output)."""
        if cnn_model_name == "custom-cnn":
            model = HipCustomCNN(common_dim)
            cnn_dim_output = common_dim
            loadModels.set_backbone_train_mode(model, backbone_train_mode)
        elif cnn_model_name in RESNET_DEPTHS:
            model = HipResNet(cnn_model_name)
            cnn_dim_output = model.num_features
            loadModels.set_backbone_train_mode(model, backbone_train_mode, last_n_layers=1)
        elif cnn_model_name == "vgg16":
            model = HipVGG16()                       # classifier[:-1] as in the reference (:79)
            cnn_dim_output = 4096
            loadModels.set_backbone_train_mode(model, backbone_train_mode, last_n_layers=1)
        elif cnn_model_name == "mobilenet-v2":
            model = HipMobileNetV2(cnn_model_name)
            cnn_dim_output = model.num_features
            loadModels.set_backbone_train_mode(model, backbone_train_mode, last_n_layers=1)
        elif cnn_model_name in ("efficientnet-b0", "efficientnet-b7"):
            model = HipEfficientNet(cnn_model_name)
            cnn_dim_output = model.num_features     # 1280 / 2560 (reference :104,110)
            loadModels.set_backbone_train_mode(model, backbone_train_mode, last_n_layers=1)
        elif cnn_model_name == "densenet169":
            model = HipDenseNet(cnn_model_name)
            cnn_dim_output = model.num_features
            if backbone_train_mode == "partial":        # reference :88-92
                for p in model.parameters():
                    p.requires_grad = False
                for p in model.features.denseblock4.parameters():
                    p.requires_grad = True
            else:
                loadModels.set_backbone_train_mode(model, backbone_train_mode)
        elif cnn_model_name.startswith(("beitv2_", "vit_", "davit_")):
            # the reference's generic timm branch (:117-152): create_model(name) + reset_classifier(0), F = num_features,
            # "partial" unfreezes the last block
            if cnn_model_name.startswith("beitv2_"):
                from hip_beit import HipBeit
                model = HipBeit(cnn_model_name)
            elif cnn_model_name.startswith("davit_"):
                from hip_davit import HipDaVit
                model = HipDaVit(cnn_model_name)
            else:
                from hip_vit import HipVisionTransformer
                model = HipVisionTransformer(cnn_model_name)
            cnn_dim_output = model.num_features
            if backbone_train_mode == "partial":
                for p in model.parameters():
                    p.requires_grad = False
                last = model.stages[-1] if hasattr(model, "stages") else model.blocks[-1]      # reference :125-131
                for p in last.parameters():
                    p.requires_grad = True
            else:
                loadModels.set_backbone_train_mode(model, backbone_train_mode)
        elif cnn_model_name in _KNOWN_WITHOUT_PLAN:
            raise NotImplementedError(
                f"Backbone '{cnn_model_name}' is accepted by the reference but has no MI355X plan yet "
                "(see DESIGN.md, scope table).")
        else:
            raise ValueError(f"Backbone '{cnn_model_name}' não implementado.")
        return model, cnn_dim_output

    @staticmethod
    def loadTextModelEncoder(text_model_encoder: str, train_mode: str = "frozen_weights"):
        """reference :162-203 -> (model, output_dim, output_dim)."""
        if text_model_encoder == "bert-base-uncased":
            from hip_bert import HipBertModel
            model = HipBertModel()                      # random init; load a HuggingFace state_dict for trained weights
            output_dim = model.config.hidden_size
            for p in model.parameters():                # reference :174-179
                p.requires_grad = train_mode == "unfrozen_weights"
            return model, output_dim, output_dim
        elif text_model_encoder == "gpt2":
            from hip_gpt2 import HipGPT2Model
            model = HipGPT2Model()
            output_dim = model.config.hidden_size
            for p in model.parameters():
                p.requires_grad = train_mode == "unfrozen_weights"
            return model, output_dim, output_dim
        elif text_model_encoder == "tab-transformer":
            categorical_cardinalities = [10] * 82
            output_dim = 85
            model = TabTransformer(categorical_cardinalities=categorical_cardinalities, num_continuous=4,
                                   output_dim=output_dim)
            return model, output_dim, output_dim
        else:
            raise ValueError(f"Text encoder '{text_model_encoder}' não suportado.")
