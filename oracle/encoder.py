"""Frame encoders (BasicEncoder), functional fp32 restatement over a state dict.

ORACLE (test infrastructure). Follows vipe/slam/networks/droid_net.py:179-232 (ResidualBlock), :290-370
(BasicEncoder: 7x7/2 stem -> norm -> relu, layer1..3 = two residual blocks each at 32 / 64 / 128 channels with stride
1 / 2 / 2, 1x1 output conv) and :510-527 (DroidNet.encode_features / encode_context: ImageNet mean / std
normalisation; the context net's 256 outputs split into tanh(net), relu(inp)).
`sd` uses the reference key layout: conv1, layer{1,2,3}.{0,1}.{conv1,conv2}, layer{2,3}.0.downsample.0, conv2,
each .weight / .bias.  norm_fn "instance" = nn.InstanceNorm2d defaults (no affine, eps 1e-5, biased variance),
"none" = identity.  Pinned by tests/golden/encoder_reference.npz (outputs of the reference classes)."""
import torch
import torch.nn.functional as F

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def _norm(x, norm_fn):
    return F.instance_norm(x, eps=1e-5) if norm_fn == "instance" else x


def _conv(sd, name, x, stride=1, pad=0):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=pad)


def _block(sd, pre, x, norm_fn, stride):
    """droid_net.py:221-232"""
    y = F.relu(_norm(_conv(sd, pre + ".conv1", x, stride, 1), norm_fn))
    y = F.relu(_norm(_conv(sd, pre + ".conv2", y, 1, 1), norm_fn))
    if stride != 1:
        x = _norm(_conv(sd, pre + ".downsample.0", x, stride, 0), norm_fn)
    return F.relu(x + y)


def encoder_forward(sd, x, norm_fn):
    """droid_net.py:352-370. x [n,3,H,W] (already normalised) -> [n,out,H/8,W/8]"""
    x = F.relu(_norm(_conv(sd, "conv1", x, 2, 3), norm_fn))
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        x = _block(sd, f"layer{li}.0", x, norm_fn, stride)
        x = _block(sd, f"layer{li}.1", x, norm_fn, 1)
    return _conv(sd, "conv2", x)


def normalize_images(images):
    """droid_net.py:512-516. images [n,3,H,W] RGB in [0,1]"""
    mean = torch.tensor(MEAN, dtype=images.dtype).view(1, 3, 1, 1)
    std = torch.tensor(STD, dtype=images.dtype).view(1, 3, 1, 1)
    return (images - mean) / std


def encode_features(sd_fnet, images):
    return encoder_forward(sd_fnet, normalize_images(images), "instance")


def encode_context(sd_cnet, images):
    out = encoder_forward(sd_cnet, normalize_images(images), "none")
    net, inp = out.split([128, 128], dim=1)
    return net.tanh(), inp.relu()
