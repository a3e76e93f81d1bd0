"""KGnet architecture tables: parameter names/shapes exactly as the reference's state_dict
(KGnet.py:125-227; 346 entries) and the layer wiring of forward_dec (KGnet.py:275-318)."""

LAYERS = [("layer1", 64, 64, 3, 1), ("layer2", 256, 128, 4, 2), ("layer3", 512, 256, 6, 2)]  # name, inplanes, planes, blocks, stride


def layers_table(layers=(3, 4, 6, 3)):
    """KGnet.py:135-137: only layers[0..2] are built (Bottleneck, expansion 4); the block counts come from the constructor
    (resnet50 [3,4,6,3], resnet101 [3,4,23,3], resnet152 [3,8,36,3], KGnet.py:377-410)."""
    return [(n, i, p, int(b), st) for (n, i, p, _, st), b in zip(LAYERS, layers[:3])]
FEAT_CH = [64, 64, 256, 512, 1024]          # c0..c4 channels
SKIP = [(64, 64, 128), (256, 64, 128), (512, 256, 512), (1024, 512, 1024)]  # skip_combine[i]: (in, out, cat)
HEADS = [("kp", 5), ("short_offset", 10), ("mid_offset", 40)]
HEAD_CH = [64, 64, 256, 512]                 # channels of c0_cat..c3_cat
EDGES = [(0, 1), (0, 2), (0, 3), (0, 4), (1, 2), (1, 3), (1, 4), (2, 3), (2, 4), (3, 4)]  # config.py:2-13


def _bn(p, c):
    return [(f"{p}.weight", (c,), "bn_w"), (f"{p}.bias", (c,), "bn_b"), (f"{p}.running_mean", (c,), "bn_rm"),
            (f"{p}.running_var", (c,), "bn_rv"), (f"{p}.num_batches_tracked", (), "bn_nbt")]


def _conv(p, cout, cin, k, bias=True):
    s = [(f"{p}.weight", (cout, cin, k, k), "conv_w")]
    if bias:
        s.append((f"{p}.bias", (cout,), "conv_b"))
    return s


def state_spec(layers=(3, 4, 6, 3)):
    """Ordered [(key, shape, kind)] == reference ResNet(Bottleneck, layers).state_dict() order."""
    s = _conv("conv1", 64, 3, 7, bias=False) + _bn("bn1", 64)
    for name, inplanes, planes, blocks, _ in layers_table(layers):
        for b in range(blocks):
            p = f"{name}.{b}"
            cin = inplanes if b == 0 else planes * 4
            s += _conv(f"{p}.conv1", planes, cin, 1, False) + _bn(f"{p}.bn1", planes)
            s += _conv(f"{p}.conv2", planes, planes, 3, False) + _bn(f"{p}.bn2", planes)
            s += _conv(f"{p}.conv3", planes * 4, planes, 1, False) + _bn(f"{p}.bn3", planes * 4)
            if b == 0:
                s += _conv(f"{p}.downsample.0", planes * 4, cin, 1, False) + _bn(f"{p}.downsample.1", planes * 4)
    s += _conv("c0_conv.0", 64, 3, 3) + _conv("c0_conv.2", 64, 64, 3)
    for i, (cin, cout, ccat) in enumerate(SKIP):
        s += _conv(f"skip_combine.{i}.up.0", cout, cin, 3) + _conv(f"skip_combine.{i}.cat_conv.0", cout, ccat, 1)
    s += _conv("seg_head.0", 64, 64, 3) + _conv("seg_head.2", 1, 64, 3)
    s += _conv("c4_up_conv.0", 512, 1024, 3) + _conv("c3_up_conv.0", 256, 512, 3)
    s += _conv("c2_up_conv.0", 64, 256, 3) + _conv("c1_up_conv.0", 64, 64, 3)
    s += _conv("c3_cat_refine.0", 512, 1024, 1) + _conv("c2_cat_refine.0", 256, 512, 1)
    s += _conv("c1_cat_refine.0", 64, 128, 1) + _conv("c0_cat_refine.0", 64, 128, 1)
    for lvl in (3, 2, 1, 0):
        c = HEAD_CH[lvl]
        for name, cout in HEADS:
            s += _conv(f"{name}_head_c{lvl}.0", c, c, 7) + _conv(f"{name}_head_c{lvl}.2", cout, c, 7)
    return s
