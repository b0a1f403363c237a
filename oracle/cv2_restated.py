"""ORACLE — test infrastructure only.  NOT part of the product path.

numpy restatement of the two OpenCV calls the reference's CGT scale-label code makes
(/root/reference/mono/model/mono_baseline/net.py:300-305 and :394-399):

    cv2.fillConvexPoly(img_zero, pts, (0, 255, 255), 1)      # 4th positional argument = lineType = 1
    cv2.cvtColor(img, cv2.COLOR_RGB2GRAY)

OpenCV (requirements.txt: `opencv-python`, version unpinned) is absent from /root/reference and from this
image, so this is **parity unpinned** third-party arithmetic: the algorithm below is the published one of
OpenCV 4.x `modules/imgproc/src/drawing.cpp` (`FillConvexPoly`, `Line`, `LineIterator`, `clipLine`) and
`color_yuv.simd.hpp` (8-bit RGB2GRAY fixed point), restated from its documented behaviour:

  * every polygon edge is first drawn with `Line(img, p0, p, color, line_type)`; lineType 1 is mapped to a
    4-connected Bresenham line (`connectivity == 1 -> 4`), drawn left-to-right, clipped to the image;
  * the interior is filled by a two-edge scan conversion in 16.16 fixed point: on row y the span is
    [(x_left + 0.5) >> 16, (x_right + 0.5) >> 16] with x advanced by a rounded per-row slope
    `dx = ((xe - xs) * 2 + (ty - y)) / (2 * (ty - y))` (C integer division, truncating);
  * RGB2GRAY: (R*4899 + G*9617 + B*1868 + 2^13) >> 14.

`tools/make_golden.py` installs this module as the `cv2` stub when it imports the real reference, and
`oracle/jp_oracle.py` uses it for `scale_label_static/dynamic`; the HIP kernel `jp_fill_convex_poly`
follows the same algorithm and is compared pixel-exactly against it.
"""
from __future__ import annotations

import numpy as np

XY_SHIFT = 16
XY_ONE = 1 << XY_SHIFT
LINE_4, LINE_8, LINE_AA = 4, 8, 16
COLOR_RGB2GRAY = 7


def _cdiv(a: int, b: int) -> int:
    """C integer division (truncation toward zero)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def clip_line(width: int, height: int, pt1, pt2):
    """cv::clipLine(Size, Point&, Point&) -> (inside?, pt1, pt2)."""
    x1, y1 = int(pt1[0]), int(pt1[1])
    x2, y2 = int(pt2[0]), int(pt2[1])
    right, bottom = width - 1, height - 1
    if width <= 0 or height <= 0:
        return False, (x1, y1), (x2, y2)
    c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8
    c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += int(float(a - y1) * (x2 - x1) / (y2 - y1))     # (int64)(double) truncation
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += int(float(a - y2) * (x2 - x1) / (y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += int(float(a - x1) * (y2 - y1) / (x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += int(float(a - x2) * (y2 - y1) / (x2 - x1))
                x2 = a
                c2 = 0
    return (c1 | c2) == 0, (x1, y1), (x2, y2)


def line_points(width: int, height: int, pt1, pt2, connectivity: int = 8):
    """cv::LineIterator(img, pt1, pt2, connectivity, leftToRight=true): the visited pixels, in order."""
    assert connectivity in (4, 8)
    x1, y1 = int(pt1[0]), int(pt1[1])
    x2, y2 = int(pt2[0]), int(pt2[1])
    if not (0 <= x1 < width and 0 <= x2 < width and 0 <= y1 < height and 0 <= y2 < height):
        ok, (x1, y1), (x2, y2) = clip_line(width, height, (x1, y1), (x2, y2))
        if not ok:
            return []
    delta_x = delta_y = 1
    dx, dy = x2 - x1, y2 - y1
    if dx < 0:                       # leftToRight: start from the left end point
        dx, dy = -dx, -dy
        x1, y1 = x2, y2
    if dy < 0:
        dy, delta_y = -dy, -1
    vert = dy > dx
    if vert:
        dx, dy = dy, dx
    if connectivity == 8:
        err, plus_delta, minus_delta, count = dx - (dy + dy), dx + dx, -(dy + dy), dx + 1
    else:
        err, plus_delta, minus_delta, count = 0, (dx + dx) + (dy + dy), -(dy + dy), dx + dy + 1
    pts = []
    x, y = x1, y1
    for _ in range(count):
        pts.append((x, y))
        mask = err < 0
        err += minus_delta + (plus_delta if mask else 0)
        # major axis step (x unless `vert`), minor axis step only when the error went negative;
        # a 4-connected line never moves along both axes in one step
        if connectivity == 8:
            major, minor = 1, (1 if mask else 0)
        else:
            major, minor = (0 if mask else 1), (1 if mask else 0)
        if vert:
            y += delta_y * major
            x += delta_x * minor
        else:
            x += delta_x * major
            y += delta_y * minor
    return pts


def _draw_line(img: np.ndarray, pt1, pt2, color, line_type):
    conn = 8 if line_type == 0 else 4 if line_type == 1 else line_type
    h, w = img.shape[:2]
    for x, y in line_points(w, h, pt1, pt2, conn):
        img[y, x] = color


def fillConvexPoly(img: np.ndarray, points, color, lineType: int = LINE_8, shift: int = 0):
    """cv2.fillConvexPoly (in place, returns img).  Only shift == 0 and non-antialiased line types are restated
    (what the reference uses)."""
    assert shift == 0 and lineType < LINE_AA, "only the reference's call pattern is restated"
    v = np.asarray(points).reshape(-1, 2).astype(np.int64)
    npts = len(v)
    if npts == 0:
        return img
    h, w = img.shape[:2]
    col = np.asarray(color[: img.shape[2]] if img.ndim == 3 else color[0]).astype(img.dtype)
    delta1 = delta2 = XY_ONE >> 1
    p0 = (int(v[-1][0]), int(v[-1][1]))
    xmin = xmax = int(v[0][0])
    ymin = ymax = int(v[0][1])
    imin = 0
    for i in range(npts):
        px, py = int(v[i][0]), int(v[i][1])
        if py < ymin:
            ymin, imin = py, i
        ymax, xmax, xmin = max(ymax, py), max(xmax, px), min(xmin, px)
        _draw_line(img, p0, (px, py), col, lineType)
        p0 = (px, py)
    if npts < 3 or xmax < 0 or ymax < 0 or xmin >= w or ymin >= h:
        return img
    ymax = min(ymax, h - 1)
    e_idx, e_di = [imin, imin], [1, npts - 1]
    e_x, e_dx, e_ye = [-XY_ONE, -XY_ONE], [0, 0], [ymin, ymin]
    edges = npts
    y = ymin
    while True:
        for i in range(2):            # line_type < CV_AA: re-target an edge whenever the scan line reaches its end
            if y >= e_ye[i]:
                idx0, di = e_idx[i], e_di[i]
                idx = idx0 + di
                if idx >= npts:
                    idx -= npts
                while True:
                    edges -= 1
                    if edges < 0:     # `for (; edges-- > 0; )` exhausted
                        break
                    ty = int(v[idx][1])
                    if ty > y:
                        xs, xe = int(v[idx0][0]) << XY_SHIFT, int(v[idx][0]) << XY_SHIFT
                        e_ye[i] = ty
                        e_dx[i] = _cdiv((xe - xs) * 2 + (ty - y), 2 * (ty - y))
                        e_x[i] = xs
                        e_idx[i] = idx
                        break
                    idx0 = idx
                    idx += di
                    if idx >= npts:
                        idx -= npts
        if edges < 0:
            break
        if y >= 0:
            left, right = (1, 0) if e_x[0] > e_x[1] else (0, 1)
            xx1 = (e_x[left] + delta1) >> XY_SHIFT
            xx2 = (e_x[right] + delta2) >> XY_SHIFT
            if xx2 >= 0 and xx1 < w:
                xx1, xx2 = max(xx1, 0), min(xx2, w - 1)
                if xx2 >= xx1:
                    img[y, xx1:xx2 + 1] = col
        e_x[0] += e_dx[0]
        e_x[1] += e_dx[1]
        y += 1
        if y > ymax:
            break
    return img


def cvtColor(img: np.ndarray, code: int):
    assert code == COLOR_RGB2GRAY and img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3
    r, g, b = (img[..., k].astype(np.int64) for k in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14).astype(np.uint8)
