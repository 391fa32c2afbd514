"""Data-side helpers next to the hot path (mirror of the reference's src/data for the functions the
keyframe selector calls into the device path)."""
