class VideoRecorder:
    """No-op recorder: golden-vector generation never renders."""

    def __init__(self, env=None, path=None, **kw):
        self.env, self.path, self.frames_per_sec = env, path, 1

    def capture_frame(self):
        pass

    def close(self):
        pass
