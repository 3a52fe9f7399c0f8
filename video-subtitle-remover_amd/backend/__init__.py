"""Mirror of the reference's ``backend`` package surface for the inpaint hot path only."""
