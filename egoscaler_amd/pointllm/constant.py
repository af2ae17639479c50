"""Token templates, same names and values as the reference's constant.py:1-25."""
IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200

DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"
SEP_TOKEN = "<sep>"

TIMESTEP_START_TOKEN = "<ts>"
TIMESTEP_END_TOKEN = "<te>"
TIMESTEP_SEP_TOKEN = "<tsep>"

COORD_X_TOKEN_TEMPLATE = "<x{p}>"
COORD_Y_TOKEN_TEMPLATE = "<y{p}>"
COORD_Z_TOKEN_TEMPLATE = "<z{p}>"
ROT_X_TOKEN_TEMPLATE = "<rx{p}>"
ROT_Y_TOKEN_TEMPLATE = "<ry{p}>"
ROT_Z_TOKEN_TEMPLATE = "<rz{p}>"

RT2_TOKEN_TEMPLATE = "<p{p}>"

DEFAULT_POINT_PATCH_TOKEN = "<point_patch>"
DEFAULT_POINT_START_TOKEN = "<point_start>"
DEFAULT_POINT_END_TOKEN = "<point_end>"
