"""The oracle pin hook: builds the REFERENCE graph -- /root/reference/model.py + module.py,
unmodified, under TensorFlow 1.x -- with the oracle's seeded weights injected into its variables, runs three
`CycleGAN.train` steps on the synthetic batch of SURVEY 8d, and dumps tests/golden/golden_tf_f32.npz in the schema of
golden_full_f32.npz (make_golden.py).  tests/test_oracle.py::test_oracle_against_tensorflow_reference then holds the
oracle to those numbers, which lifts "parity unpinned".

TensorFlow 1.x (the reference needs tf.contrib, i.e. TF <= 1.15) is NOT installable in the build container (no network,
no wheel), so this script has never been run there; it is written against the public TF1 API and the reference's own
variable names, and is meant for anyone who has such an environment:

    pip install 'tensorflow<2'        # python <= 3.7
    python tests/golden/make_tf_golden.py /path/to/Voice-Converter-CycleGAN

The only thing it takes from this repo is the weight initialiser and the input generator (seeded numpy); every number it
writes comes out of TensorFlow.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
LOSSES = ('cycle_loss', 'identity_loss', 'generator_loss_A2B', 'generator_loss_B2A', 'generator_loss',
          'discriminator_loss_A', 'discriminator_loss_B', 'discriminator_loss')


def probe(a, i):
    r = np.random.default_rng(1000 + i).standard_normal(a.size)
    return float(np.asarray(a).reshape(-1).astype(np.float64) @ r)


def main(ref_dir):
    import tensorflow as tf
    assert tf.__version__.startswith('1.'), 'the reference uses tf.contrib / tf.layers: TensorFlow 1.x needed'
    from oracle import cyclegan_oracle as O                       # seeded weights + inputs only
    sys.path.insert(0, ref_dir)
    from model import CycleGAN                                    # the reference's model.py (imports its module.py, utils.py)
    P = O.init_params(seed=42, randomize_affine=True)
    A, B = O.synthetic_batch(1, 24, 128)
    model = CycleGAN(num_features=24, log_dir=os.path.join(HERE, '_tf_log'))
    # inject the oracle's weights: the oracle keys ARE the TF variable names (without ':0')
    tvars = {v.name[:-2]: v for v in tf.trainable_variables()}
    missing = sorted(set(P) - set(tvars)), sorted(set(tvars) - set(P))
    assert not missing[0] and not missing[1], 'variable name mismatch: %s' % (missing,)
    model.sess.run([tf.assign(tvars[k], P[k].reshape(tvars[k].shape.as_list())) for k in sorted(P)])
    # gradients of step 0 w.r.t. every variable, from the reference's own loss tensors (model.py:72, 90)
    g_vars = [tvars[k] for k in sorted(P) if 'generator' in k]
    d_vars = [tvars[k] for k in sorted(P) if 'discriminator' in k]
    g_grads = tf.gradients(model.generator_loss, g_vars)
    d_grads = tf.gradients(model.discriminator_loss, d_vars)
    feed_g = {model.lambda_cycle: 10.0, model.lambda_identity: 5.0, model.input_A_real: A, model.input_B_real: B}
    gen_A, gen_B, gg = model.sess.run([model.generation_A, model.generation_B, g_grads], feed_dict=feed_g)
    dg = model.sess.run(d_grads, feed_dict={model.input_A_real: A, model.input_B_real: B,
                                            model.input_A_fake: gen_A, model.input_B_fake: gen_B})
    grads = {v.name[:-2]: g for v, g in list(zip(g_vars, gg)) + list(zip(d_vars, dg))}
    names = sorted(grads)
    out = {'gen_A': gen_A, 'gen_B': gen_B, 'grad_names': np.array(names),
           'grad_l2': np.array([np.linalg.norm(grads[n].astype(np.float64)) for n in names]),
           'grad_proj': np.array([probe(grads[n], i) for i, n in enumerate(names)]),
           'tf_version': np.array(tf.__version__)}
    # three train steps through the reference's own train() (model.py:110-125); the 8 scalars are re-evaluated on the
    # pre-update weights exactly as its summaries are (fetched in the same sess.run as the optimizer op)
    fetch_g = [model.cycle_loss, model.identity_loss, model.generator_loss_A2B, model.generator_loss_B2A, model.generator_loss]
    fetch_d = [model.discriminator_loss_A, model.discriminator_loss_B, model.discriminator_loss]
    for step in range(3):
        sg = model.sess.run(fetch_g, feed_dict=feed_g)
        fa, fb = model.sess.run([model.generation_A, model.generation_B], feed_dict=feed_g)
        sd = model.sess.run(fetch_d, feed_dict={model.input_A_real: A, model.input_B_real: B,
                                                model.input_A_fake: fa, model.input_B_fake: fb})
        out['step%d/losses' % step] = np.array(sg + sd, np.float32)
        model.train(input_A=A, input_B=B, lambda_cycle=10.0, lambda_identity=5.0,
                    generator_learning_rate=2e-4, discriminator_learning_rate=1e-4)
    # the parameters after three TF-Adam steps, pinned like the gradients
    final = {v.name[:-2]: model.sess.run(v) for v in tf.trainable_variables()}
    out['param_l2'] = np.array([np.linalg.norm(final[n].astype(np.float64)) for n in names])
    out['param_proj'] = np.array([probe(final[n], i) for i, n in enumerate(names)])
    np.savez_compressed(os.path.join(HERE, 'golden_tf_f32.npz'), **out)
    print('wrote golden_tf_f32.npz (TensorFlow %s)' % tf.__version__)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else '/root/reference')
