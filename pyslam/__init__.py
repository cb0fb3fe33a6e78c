"""Drop-in module names of the reference package (``pyslam.problem``,
``pyslam.residuals``, ``pyslam.losses``, ``pyslam.sensors``, ``pyslam.utils``,
``pyslam.pipelines.ransac``, ``pyslam.metrics``), all backed by pyslam_amd.  The
reference's other pipelines / visualizers (cv2, viso2, matplotlib front-ends) are
out of scope -- see DESIGN.md."""
