"""Mirrors of the reference's ``utils/`` pieces that sit on the plane-sweep path's training loss."""
