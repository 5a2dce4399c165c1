"""Only the device-side target encoding of the reference's dataset classes lives here;
image decoding and augmentation stay outside this build."""
from .targets import encode_targets   # noqa: F401
