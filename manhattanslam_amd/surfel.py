"""Host mirror of SurfelFusion (reference include/SurfelFusion.h:43-139) -- filled in with the kernels."""


class SurfelFusion:
    pass


class SurfelMap:
    pass
