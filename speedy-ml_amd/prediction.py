"""The start of a prediction (src/mod_reservoir.f90:791-961) for the reservoirs resident in a bank: the host-side orchestration around
`synchronize` (device: ReservoirBank.synchronize, one launch per column for all reservoirs).

    initialize_prediction (:791-938): saved_state = 0; synchronize over un_noisy_sync / timestep - 1 = 359 columns of the
                                      timestep-strided prediction data (un_noisy_sync = 2160 h, :818-824)
    start_prediction      (:940-961): synchronize_print over synclength / timestep - 1 columns; then
                                      feedback    = predictiondata(:, synclength / timestep)
                                      local_model = imperfect_model_states(:, synclength / timestep + 1)
`predictiondata` / `imperfect_model_states` are what get_prediction_data (:605-789) builds: the standardised input vectors and the
standardised SPEEDY forecasts at every `timestep`-th hour of the window (columns are 1-based in the reference, 0-based here)."""
import numpy as np

UN_NOISY_SYNC = 2160      # src/mod_reservoir.f90:818


def _columns_to_device(bank, per_slot, ncols):
    """[ncols][capacity][max_d] device tensor from per-slot (d, >= ncols) host arrays (None = slot not driven)"""
    import torch
    buf = np.zeros((ncols, bank.capacity, bank.max_d))
    for slot, data in enumerate(per_slot):
        if data is None:
            continue
        data = np.asarray(data, dtype=np.float64)
        assert data.shape[1] >= ncols, "prediction data shorter than the synchronisation window"
        buf[:, slot, :data.shape[0]] = data[:, :ncols].T
    return torch.from_numpy(buf).cuda()


def initialize_prediction(bank, predictiondata, timestep, un_noisy_sync=UN_NOISY_SYNC, stream=None):
    """States to zero, then synchronize over un_noisy_sync / timestep - 1 columns (:816-824).  predictiondata: per-slot (d, L) arrays."""
    ncols = un_noisy_sync // timestep - 1
    for slot, data in enumerate(predictiondata):
        if data is not None:
            bank.set_state(slot, np.zeros(len(bank.get_state(slot))))
    dev = _columns_to_device(bank, predictiondata, ncols)
    bank.synchronize(dev.data_ptr(), ncols, stream=stream)
    return ncols


def start_prediction(bank, predictiondata, imperfect_model_states, synclength, timestep, stream=None):
    """synchronize_print over synclength / timestep - 1 columns, continuing from the states left by initialize_prediction, then the
    first feedback / local_model of the forecast (:949-960).  Returns the number of columns consumed."""
    ncols = synclength // timestep - 1
    dev = _columns_to_device(bank, predictiondata, ncols)
    bank.synchronize(dev.data_ptr(), ncols, stream=stream)
    for slot, data in enumerate(predictiondata):
        if data is None:
            continue
        bank.set_feedback(slot, np.ascontiguousarray(np.asarray(data)[:, ncols]))                 # column synclength/timestep (1-based)
        if imperfect_model_states is not None and imperfect_model_states[slot] is not None:
            bank.set_local_model(slot, np.ascontiguousarray(np.asarray(imperfect_model_states[slot])[:, ncols + 1]))
    return ncols

