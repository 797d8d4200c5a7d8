"""Mirror of the reference's ``evaluation`` package for the step that follows the plane-sweep path
(SURVEY section 8f-3): the geometric-consistency filter.  Only ``filtering`` is provided; the rest of the
reference's evaluation pipeline (COLMAP / fusibile glue, metrics) is out of scope and keeps calling it."""
