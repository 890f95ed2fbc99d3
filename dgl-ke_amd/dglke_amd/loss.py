"""`LossGenerator` with the reference's interface (models/pytorch/loss.py:41-98,
models/base_loss.py:61-160), computed by the fused HIP loss kernel (forward values and
d loss / d score in one pass; the softmax of the adversarial weighting is detached exactly like
loss.py:88)."""
import torch as th

from . import ops


class LossGenerator(object):
    def __init__(self, args, loss_genre='Logsigmoid', neg_adversarial_sampling=False,
                 adversarial_temperature=1.0, pairwise=False):
        if loss_genre not in ('Hinge', 'Logistic', 'Logsigmoid', 'BCE'):
            raise ValueError('loss genre %s is not support' % loss_genre)
        if pairwise and neg_adversarial_sampling:
            raise ValueError('loss cannot be pairwise and adversarial sampled')
        if pairwise and loss_genre not in ['Logistic', 'Hinge']:
            raise ValueError('{} loss cannot be applied to pairwise loss function'.format(loss_genre))
        self.loss_genre = loss_genre
        self.pairwise = pairwise
        self.neg_adversarial_sampling = neg_adversarial_sampling
        self.adversarial_temperature = adversarial_temperature if neg_adversarial_sampling else 0
        self.margin = getattr(args, 'margin', 1.0) if args is not None else 1.0
        if self.margin is None:
            self.margin = 1.0
        self.neg_label = 0 if loss_genre == 'BCE' else -1

    def get_total_loss(self, pos_score, neg_score, edge_weight=None):
        """returns (loss, log) like loss.py:69-98.  The three `.item()` host syncs of the
        reference (loss.py:95-97) are replaced by ONE 12-byte device-to-host copy."""
        if edge_weight is not None:
            edge_weight = edge_weight.reshape(-1)
        loss, loss3 = ops.loss_fwd_bwd(pos_score, neg_score, edge_weight, self.loss_genre,
                                       self.neg_adversarial_sampling,
                                       self.adversarial_temperature, self.pairwise, self.margin)
        vals = loss3.tolist()
        if self.pairwise:
            return loss, {'loss': vals[2]}
        return loss, {'pos_loss': vals[0], 'neg_loss': vals[1], 'loss': vals[2]}
