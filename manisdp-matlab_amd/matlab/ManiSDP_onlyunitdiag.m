function [X, obj, data] = ManiSDP_onlyunitdiag(C, options)
%MANISDP_ONLYUNITDIAG  GPU drop-in for the reference's src/primal/ManiSDP_onlyunitdiag.m:
%   Min <C, X>  s.t.  X >= 0,  X_ii = 1.   Same call, option names, defaults, printed lines and data fields;
%   the work is done by msdp_al_engine over libmanisdp_hip (see that file).
if nargin < 2, options = struct(); end
defaults = {'p0', 2; 'AL_maxiter', 20; 'tol', 1e-8; 'theta', 1e-1; 'delta', 8; 'alpha', 0.5; ...
            'tolgradnorm', 1e-8; 'TR_maxinner', 100; 'TR_maxiter', 40; 'line_search', 0};
[X, obj, data] = msdp_al_engine('onlyunitdiag', struct('n', size(C, 1), 'C', C), options, defaults);
end
