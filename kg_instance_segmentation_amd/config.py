"""Constants of the reference's config.py:2-28 (keypoint graph of a box: 4 corners + centre)."""
EDGES = [(0, 1), (0, 2), (0, 3), (0, 4), (1, 2), (1, 3), (1, 4), (2, 3), (2, 4), (3, 4)]
NUM_EDGES = len(EDGES)
NUM_KPS = 5
KP_RADIUS = 5
KEYPOINTS = ["tl", "tr", "bl", "br", "center"]
