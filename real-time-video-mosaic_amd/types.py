"""Plain-Python stand-ins for the two OpenCV value types the VideMosaic API hands out."""


class KeyPoint:
    """cv2.KeyPoint look-alike: pt (x, y) in level-0 pixels, size, angle (degrees),
    response, octave, class_id."""
    __slots__ = ("pt", "size", "angle", "response", "octave", "class_id")

    def __init__(self, x, y, size, angle=-1.0, response=0.0, octave=0, class_id=-1):
        self.pt = (float(x), float(y))
        self.size = float(size)
        self.angle = float(angle)
        self.response = float(response)
        self.octave = int(octave)
        self.class_id = int(class_id)

    def __repr__(self):
        return "KeyPoint(pt=(%.2f, %.2f), size=%.1f, angle=%.2f, octave=%d)" % (
            self.pt[0], self.pt[1], self.size, self.angle, self.octave)


class DMatch:
    """cv2.DMatch look-alike."""
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, queryIdx, trainIdx, distance, imgIdx=0):
        self.queryIdx = int(queryIdx)
        self.trainIdx = int(trainIdx)
        self.imgIdx = int(imgIdx)
        self.distance = float(distance)

    def __repr__(self):
        return "DMatch(%d -> %d, %.1f)" % (self.queryIdx, self.trainIdx, self.distance)


def keypoints_from_arrays(f):
    """list[KeyPoint] from the dict of arrays the device hands back."""
    xy, size, angle, resp, octv = f["xy"], f["size"], f["angle"], f["response"], f["octave"]
    return [KeyPoint(xy[i, 0], xy[i, 1], size[i], angle[i], resp[i], octv[i]) for i in range(len(xy))]


def matches_from_arrays(qi, ti, di):
    return [DMatch(q, t, d) for q, t, d in zip(qi.tolist(), ti.tolist(), di.tolist())]
