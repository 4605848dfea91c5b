"""CPU oracle for the MultimodalModel hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(``multimodal-model-skin-lesion-classifier_amd/``) may import from here; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` do, and only as the checker / the timed CPU baseline.

What it is: a plain PyTorch-CPU fp32 restatement of the reference's algorithm
for ``MultimodalModel.forward`` (image backbone -> projections -> L=1
multi-head attention -> fusion dispatch -> classifier head), i.e. of

  /root/reference/src/scripts/benchmark/models/multimodalIntraInterModal.py:13-416
  /root/reference/src/scripts/benchmark/models/loadImageModelClassifier.py:15-203
  /root/reference/src/scripts/benchmark/models/gatedResidualBlock.py:4-17
  /root/reference/src/scripts/benchmark/models/metablock.py:4-32
  /root/reference/src/scripts/benchmark/models/tab_transformer.py:6-60

Pinning status
--------------
* Head (projections, 4x MHA, all 18 fusion strings, MetaBlock, gated residual
  block, TabTransformer class): PINNED.  ``oracle/gen_golden.py`` imports the
  real reference from /root/reference in the build container (the reference has
  no tests or golden vectors of its own) and writes logits / loss / gradient /
  post-Adam fixtures to ``tests/golden/``; ``tests/test_oracle_golden.py``
  checks this restatement against them.
* Backbones (ResNet-18/50, ...): PARITY UNPINNED against the reference's
  third-party dependency (torchvision==0.19.1, requirements.txt:4), which is
  neither vendored under /root/reference nor installed here.  The restatement
  follows the published torchvision v1.5 architecture (module names, strides,
  eps/momentum, init) and is built from ``torch.nn.functional`` CPU ops only.
"""
