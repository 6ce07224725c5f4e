"""spectral_amd: MI355X-native batched Bezier-in-corridor trajectory QP solver.

Drop-in for the hot path of Srujan-D/spectral (find_traj -> Optimize) behind the
reference's own C-ABI; see DESIGN.md.
"""
