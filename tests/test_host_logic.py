"""Host-side mirrors of the reference's config / VarBuilder surface and the sharding helper (no GPU)."""
import numpy as np
import pytest


def test_configs_mirror_reference():
    import candle_birefnet_amd as cb
    c = cb.BiRefNetConfig.swin_l()
    assert c.lateral_channels() == [384, 768, 1536, 3072] and c.x4_channels() == 5760
    assert cb.SwinConfig.swin_t().window_size == 7 and cb.SwinConfig.swin_b().embed_dim == 128
    assert cb.SwinConfig.swin_l().stage_channels() == [192, 384, 768, 1536]
    cc = c.to_c()
    assert list(cc.depths) == [2, 2, 18, 2] and cc.deform_mode == 0
    c.deform_mode = "nope"
    with pytest.raises(ValueError):
        c.to_c()


def test_weight_spec_matches_survey_appendix_a():
    import candle_birefnet_amd as cb
    spec = cb.birefnet_weight_spec(cb.BiRefNetConfig())
    names = {n for n, _, _ in spec}
    assert len(names) == len(spec)
    total = sum(int(np.prod(s)) for _, s, _ in spec)
    assert abs(total / 1e6 - 220.2) < 0.05               # 220.2 M parameters
    for must in ("bb.patch_embed.proj.weight", "bb.layers.2.blocks.17.attn.relative_position_bias_table", "bb.layers.2.downsample.reduction.weight",
                 "bb.norm3.bias", "squeeze_module.0.dec_att.aspp_deforms.2.atrous_conv.offset_conv.weight", "decoder.gdt_convs_pred_4.0.weight",
                 "decoder.conv_ms_spvn_2.bias", "decoder.conv_out1.0.weight", "decoder.lateral_block4.conv.weight",
                 "decoder.decoder_block1.dec_att.global_avg_pool.2.running_var"):
        assert must in names, must
    shapes = {n: s for n, s, _ in spec}
    assert shapes["decoder.conv_out1.0.weight"] == (1, 240, 1, 1) and shapes["squeeze_module.0.conv_in.weight"] == (64, 5760, 3, 3)
    assert shapes["bb.layers.3.blocks.0.attn.relative_position_bias_table"] == (529, 48)
    assert shapes["decoder.decoder_block1.dec_att.aspp_deforms.2.atrous_conv.offset_conv.weight"] == (98, 64, 7, 7)


def test_synth_is_deterministic_and_nonzero_bias_table():
    import candle_birefnet_amd as cb
    a = cb.weights.synth_tensor("bb.layers.0.blocks.0.attn.relative_position_bias_table", (529, 6), "rel_bias", 42)
    b = cb.weights.synth_tensor("bb.layers.0.blocks.0.attn.relative_position_bias_table", (529, 6), "rel_bias", 42)
    assert np.array_equal(a, b) and np.abs(a).mean() > 0.2     # the table must be non-zero
    x0, x1 = cb.synth_input(2, 8, 8), cb.synth_input(1, 8, 8, seed0=1001)
    assert np.array_equal(x0[1], x1[0])                          # image b always comes from seed 1000 + b


def test_varbuilder_errors():
    import candle_birefnet_amd as cb
    vb = cb.VarBuilder.from_tensors({"a.b.weight": np.zeros((2, 3), np.float32)})
    assert vb.pp("a").pp("b").get((2, 3), "weight").shape == (2, 3)
    with pytest.raises(KeyError):
        vb.pp("a").get((2, 3), "nope")
    with pytest.raises(ValueError):
        vb.pp("a").pp("b").get((3, 2), "weight")


def test_shard_range_partitions():
    from candle_birefnet_amd.shard import shard_range
    for gb in (0, 1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            r = [shard_range(gb, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == gb
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)
