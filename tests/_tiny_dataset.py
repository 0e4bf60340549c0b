"""A tiny RSVQA-LR-shaped dataset written with the reference's file names (vqa_model/train.py:260-271,
test.py:136-143): all_questions.json / all_answers.json and LR_split_{train,val,test}_{questions,answers,images}.json,
in the JSON layout VocabEncoder.py:24-37 and VQALoader.py:62-70, 128-139 read (lists indexed by id, an ``active`` flag
per entry, ``questions_ids`` per image, ``answers_ids`` per question), plus one precomputed feature file per image id.
Test data, generated from a seed; nothing here comes from the reference's own data (it ships none)."""
import json
import os

import numpy as np

OBJECTS = ['road', 'building', 'forest', 'river', 'field', 'lake', 'farm', 'bridge']
TYPES = ('presence', 'count', 'comp', 'rural_urban')


def write_lr_dataset(root, n_images=24, R=9, Dv=32, seed=0, stage_maps=None, splits=(('train', 0.6), ('val', 0.2), ('test', 0.2))):
    """Returns {'data_path', 'features_dir', 'images': {split: [ids]}, 'n_questions': {split: n}}.
    stage_maps: ((C_k, grid_k), ...) writes .npz stage maps (VQAGAPModel) instead of (R, Dv) region features."""
    rng = np.random.default_rng(seed)
    root = str(root)
    feats = os.path.join(root, 'feats')
    os.makedirs(feats, exist_ok=True)
    questions, answers, images = [], [], []
    for img in range(n_images):
        qids = []
        for _ in range(int(rng.integers(2, 6))):
            t = TYPES[int(rng.integers(0, 4))]
            a, b = (OBJECTS[int(i)] for i in rng.integers(0, len(OBJECTS), 2))
            if t == 'presence':
                text, ans = f'Is there a {a} next to the {b}?', ('yes' if rng.random() < 0.5 else 'no')
            elif t == 'count':
                text, ans = f'How many {a}s are there in the image?', str(int(rng.integers(0, 1500)))
            elif t == 'comp':
                text, ans = f'Are there more {a}s than {b}s.', ('yes' if rng.random() < 0.5 else 'no')
            else:
                text, ans = 'Is it a rural or an urban area', ('rural' if rng.random() < 0.5 else 'urban')
            qid = len(questions)
            questions.append({'id': qid, 'question': text, 'type': t, 'img_id': img, 'answers_ids': [qid], 'active': True})
            answers.append({'id': qid, 'answer': ans, 'question_id': qid, 'active': True})
            qids.append(qid)
        images.append({'id': img, 'active': True, 'questions_ids': qids})
        if stage_maps:
            np.savez(os.path.join(feats, f'{img}.npz'),
                     **{f's{k}': np.maximum(rng.standard_normal((c, g, g)), 0).astype(np.float32)
                        for k, (c, g) in enumerate(stage_maps)})
        else:
            np.save(os.path.join(feats, f'{img}.npy'), np.maximum(rng.standard_normal((R, Dv)), 0).astype(np.float32))

    def dump(name, key, entries):
        with open(os.path.join(root, name), 'w') as fh:
            json.dump({key: entries}, fh)

    dump('all_questions.json', 'questions', questions)
    dump('all_answers.json', 'answers', answers)
    out = {'data_path': root, 'features_dir': feats, 'images': {}, 'n_questions': {}}
    start = 0
    for si, (split, share) in enumerate(splits):
        stop = n_images if si + 1 == len(splits) else start + int(round(n_images * share))
        mine = set(range(start, stop))
        start = stop
        # the split files keep every entry (ids are list positions) and switch the other splits' images off
        dump(f'LR_split_{split}_images.json', 'images', [dict(im, active=im['id'] in mine) for im in images])
        dump(f'LR_split_{split}_questions.json', 'questions', [dict(q, active=q['img_id'] in mine) for q in questions])
        dump(f'LR_split_{split}_answers.json', 'answers', [dict(a, active=questions[a['id']]['img_id'] in mine) for a in answers])
        out['images'][split] = sorted(mine)
        out['n_questions'][split] = sum(len(images[i]['questions_ids']) for i in mine)
    return out
