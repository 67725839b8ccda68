"""Canonical parameter order of the C ABI (include/vsrlab_hip.h), by reference state_dict key."""


def spynet_keys(prefix=""):
    keys = []
    for lvl in range(6):
        for j in range(5):
            keys += [f"{prefix}basic_module.{lvl}.basic_module.{j}.conv.0.weight",
                     f"{prefix}basic_module.{lvl}.basic_module.{j}.conv.0.bias"]
    return keys + [prefix + "mean", prefix + "std"]


def basicvsr_keys(res_blocks, upscale=4):
    """(keys, n_trainable): the first n_trainable keys are the non-SPyNet tensors."""
    keys = []
    for trunk in ("backward_resblocks", "forward_resblocks"):
        keys += [f"{trunk}.conv.0.weight", f"{trunk}.conv.0.bias"]
        for i in range(res_blocks):
            for j in (1, 2):
                keys += [f"{trunk}.res_block.{i}.conv{j}.weight", f"{trunk}.res_block.{i}.conv{j}.bias"]
    keys += ["point_conv.0.weight", "point_conv.0.bias"]
    for k in range(upscale // 2):
        keys += [f"upsample.{k}.upconv.weight", f"upsample.{k}.upconv.bias"]
    keys += ["conv_last.0.weight", "conv_last.0.bias", "conv_last.2.weight", "conv_last.2.bias"]
    n_trainable = len(keys)
    return keys + spynet_keys("spynet."), n_trainable
