"""Inert stand-in so that `/root/reference/sam2/sam2/utils/transforms.py:12` imports.

Container-only test infrastructure (used by oracle/ref_import.py when generating
golden vectors); never imported by the product package.  Only the three names the
reference imports exist.  At the 1024x1024 inputs used for every golden vector
`Resize` is the identity, `Normalize` is (x - mean) / std.
"""
