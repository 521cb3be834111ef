"""Minimal JointInfo (the reference takes posepile.joint_info.JointInfo, multiperson_model.py:4,25):
names, stick-figure edges and the left/right mirror mapping used to un-swap flipped TTA crops
(multiperson_model.py:249-251)."""
import numpy as np


class JointInfo:
    def __init__(self, joints, edges):
        self.names = [str(n) for n in joints]
        self.n_joints = len(self.names)
        self.stick_figure_edges = [(int(a), int(b)) for a, b in edges]
        index = {n: i for i, n in enumerate(self.names)}
        mapping = []
        for n in self.names:
            # posepile convention: a leading 'l'/'r' marks the side
            other = ('r' + n[1:]) if n.startswith('l') else ('l' + n[1:]) if n.startswith('r') else n
            mapping.append(index.get(other, index[n]))
        self.mirror_mapping = np.array(mapping, dtype=np.int64)
