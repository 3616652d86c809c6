"""Host-side mirror of the reference's ``dlib`` surface for the SR hot path
(``dlib.models``, ``dlib.loss`` / ``dlib.losses``, ``dlib.metrics``,
``dlib.learning``, ``dlib.utils``), backed by libsrhip.  Unlike the reference's
``dlib/__init__.py`` (which pulls the WSOL/segmentation legacy and needs
torchvision) nothing heavy is imported here."""
