"""Host-side mirror of the obstacle-mask part of the reference's analyze_for_navigation
(main.py:1051-1232) and of find_path_astar's occupancy matrix (main.py:1344-1359).

The per-pixel work runs in librtvm.so (`rtvm_nav_obstacles`); what stays here is the
arithmetic on the list of detections (main.py:1074-1104), which is a few dozen integers.
Drawing, contour extraction and the A* search itself are not part of this package.
"""
import numpy as np

from .mosaic import _stage_ctx

DANGER_CLASSES = ["fire", "smoke"]
VEHICLE_CLASSES = ["car", "truck", "bus", "motorcycle"]
LIVING_CLASSES = ["person", "dog", "horse", "cat", "bird", "cow", "sheep"]
STATIC_CLASSES = ["bicycle", "building"]


def obstacle_boxes(detections, height, width):
    """main.py:1079-1107: the rectangle cv2.rectangle fills for every detection of a known
    class, grown by a buffer that depends on the class and the box area.  Returns
    (boxes int32 [n, 4] as x1, y1, x2, y2 with both corners included, weights float32 [n])."""
    boxes, weights = [], []
    for det in detections:
        cls = det["class"]
        if cls not in DANGER_CLASSES + VEHICLE_CLASSES + LIVING_CLASSES + STATIC_CLASSES:
            continue
        x1, y1, x2, y2 = det["box"]
        area = (x2 - x1) * (y2 - y1)
        size_factor = min(1.5, max(1.0, area / 10000.0))
        if cls in DANGER_CLASSES:
            buffer, weight = int(40 * size_factor), 1.0
        elif cls in VEHICLE_CLASSES:
            buffer, weight = int(25 * size_factor), 0.9
        elif cls in LIVING_CLASSES:
            buffer, weight = int(20 * size_factor), 0.85
        else:
            buffer, weight = int(15 * size_factor), 0.7
        boxes.append((max(0, x1 - buffer), max(0, y1 - buffer), min(width, x2 + buffer), min(height, y2 + buffer)))
        weights.append(weight)
    return np.asarray(boxes, np.int32).reshape(-1, 4), np.asarray(weights, np.float32)


def navigation_obstacles(frame, detections=(), grid_scale=None, ctx=None):
    """The masks analyze_for_navigation builds from `frame` (BGR u8) and the detections
    (dicts with 'class' and 'box'), main.py:1065-1232: dict with mask_fire, mask_smoke,
    texture_mask (the input of findContours, main.py:1216), obstacles (main.py:1228),
    obstacles_nav (obstacles_for_navigation, main.py:1232) and, with grid_scale, `grid`
    (the matrix of find_path_astar for obstacles_nav)."""
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    boxes, _ = obstacle_boxes(detections, frame.shape[0], frame.shape[1])
    return (ctx or _stage_ctx()).nav_obstacles(frame, boxes, grid_scale)


def canny_edges(gray, threshold1, threshold2, ctx=None):
    """cv2.Canny(gray, threshold1, threshold2) as main.py:489 and main.py:1141 call it."""
    return (ctx or _stage_ctx()).canny(gray, threshold1, threshold2)


def occupancy_grid(obstacle_mask, scale=4, ctx=None):
    """main.py:1346-1359: `matrix` of find_path_astar as a u8 array [h // scale, w // scale]."""
    return (ctx or _stage_ctx()).occupancy_grid(obstacle_mask, scale)
