"""Secondary pin of the detectron2 half of the oracle ("parity unpinned" -- detectron2 is absent from the
reference tree and from this image).  NOT the reference: an independent implementation that happens to be installed,
transformers' `ResNetModel`, configured like detectron2's R50 (bottleneck blocks [3, 4, 6, 3], stride in the first 1x1 =
STRIDE_IN_1X1, 7x7/2 stem + 3x3/2 max-pool, BatchNorm in eval mode = FrozenBN with eps 1e-5) and loaded with the same
weights, must produce the oracle's res2..res5 stage outputs.  It pins the block structure, stride placement, padding,
residual/ReLU order and the FrozenBN arithmetic of `oracle/d2_rcnn.resnet_fpn`; FPN, RPN and ROI heads stay unpinned."""
import pytest
import torch

transformers = pytest.importorskip("transformers")


def _hf_resnet_from_d2(sd):
    from transformers import ResNetConfig, ResNetModel
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3], layer_type="bottleneck",
                       hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=True)
    m = ResNetModel(cfg).eval()
    new = {}
    bu = "backbone.bottom_up."

    def put(dst, src):
        new[dst + ".convolution.weight"] = sd[src + ".weight"]
        for a, b in (("weight", "weight"), ("bias", "bias"), ("running_mean", "running_mean"), ("running_var", "running_var")):
            new[f"{dst}.normalization.{a}"] = sd[f"{src}.norm.{b}"]
    put("embedder.embedder", bu + "stem.conv1")
    for si, nb in enumerate((3, 4, 6, 3)):
        for b in range(nb):
            d, s_ = f"encoder.stages.{si}.layers.{b}", f"{bu}res{si + 2}.{b}"
            if b == 0:
                put(d + ".shortcut", s_ + ".shortcut")
            for j, name in enumerate(("conv1", "conv2", "conv3")):
                put(f"{d}.layer.{j}", f"{s_}.{name}")
    own = m.state_dict()
    missing = [k for k in own if k not in new and not k.endswith("num_batches_tracked")]
    assert not missing, missing[:5]
    m.load_state_dict({**{k: v for k, v in own.items() if k.endswith("num_batches_tracked")}, **new})
    return m


@pytest.mark.parametrize("hw", [(96, 128), (71, 93)])
def test_r50_stage_outputs_match_an_independent_resnet(hw):
    from aldi_amd import synthetic as syn
    from oracle import d2_rcnn as d2
    sd = syn.init_state_dict(8, seed=1)
    cfg = d2.make_cfg(num_classes=8)
    g = torch.Generator().manual_seed(3)
    img = torch.randint(0, 256, (3, hw[0], hw[1]), generator=g, dtype=torch.uint8)
    x, _ = d2.preprocess(cfg, [img])
    stages = []
    with torch.no_grad():
        d2.resnet_fpn(cfg, sd, x, stages_out=stages)
        hf = _hf_resnet_from_d2(sd)
        out = hf(pixel_values=x, output_hidden_states=True)
    hs = out.hidden_states[1:]                      # after each of the four stages
    assert len(hs) == 4 == len(stages)
    for i, (a, b) in enumerate(zip(stages, hs)):
        assert a.shape == b.shape, (i, a.shape, b.shape)
        assert float(a.abs().max()) > 0
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max())), (i, float((a - b).abs().max()))
