"""ORACLE — test infrastructure only.  NOT part of the product path.

torchvision is absent from the image and from /root/reference (requirements.txt: `torchvision>=0.4.0`, unpinned):
`transforms.ColorJitter` on float tensors restated from torchvision.transforms._functional_tensor's published algorithm
(adjust_brightness / adjust_contrast / adjust_saturation / adjust_hue, _blend, rgb_to_grayscale, _rgb2hsv, _hsv2rgb).
Parity unpinned (third party); the HIP kernel jp_color_jitter_op is compared against this restatement."""
import torch


def _gray(img):
    r, g, b = img.unbind(-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(-3)


def _blend(a, b, f):
    return (f * a + (1.0 - f) * b).clamp(0, 1)


def adjust_brightness(img, f):
    return _blend(img, torch.zeros_like(img), f)


def adjust_contrast(img, f):
    return _blend(img, _gray(img).mean(dim=(-3, -2, -1), keepdim=True), f)


def adjust_saturation(img, f):
    return _blend(img, _gray(img), f)


def _rgb2hsv(img):
    r, g, b = img.unbind(-3)
    maxc, minc = torch.max(img, dim=-3).values, torch.min(img, dim=-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    crd = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return torch.stack((h, s, maxc), dim=-3)


def _hsv2rgb(img):
    h, s, v = img.unbind(-3)
    i = torch.floor(h * 6.0)
    f = (h * 6.0) - i
    i = i.to(torch.int32) % 6
    p = torch.clamp(v * (1.0 - s), 0.0, 1.0)
    q = torch.clamp(v * (1.0 - s * f), 0.0, 1.0)
    t = torch.clamp(v * (1.0 - (s * (1.0 - f))), 0.0, 1.0)
    mask = i.unsqueeze(-3) == torch.arange(6).view(-1, 1, 1)
    a1 = torch.stack((v, q, p, p, t, v), dim=-3)
    a2 = torch.stack((t, v, v, q, p, p), dim=-3)
    a3 = torch.stack((p, p, t, v, v, q), dim=-3)
    a4 = torch.stack((a1, a2, a3), dim=-4)
    return torch.einsum("...ijk, ...xijk -> ...xjk", mask.to(img.dtype), a4)


def adjust_hue(img, f):
    hsv = _rgb2hsv(img)
    h, s, v = hsv.unbind(-3)
    h = (h + f) % 1.0
    return _hsv2rgb(torch.stack((h, s, v), dim=-3))


OPS = [adjust_brightness, adjust_contrast, adjust_saturation, adjust_hue]


def color_jitter(img, order, factors):
    for op in order:
        img = OPS[op](img, factors[op])
    return img
