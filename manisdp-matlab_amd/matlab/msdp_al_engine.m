function [X, obj, data] = msdp_al_engine(kind, prob, options, defaults)
%MSDP_AL_ENGINE  Outer loop shared by the GPU drop-ins ManiSDP_onlyunitdiag / ManiSDP_unitdiag /
%   ManiSDP_unittrace / ManiSDP of this directory.
%
%   The four reference entry points (src/primal/ManiSDP_onlyunitdiag.m, ManiSDP_unitdiag.m,
%   ManiSDP_unittrace.m, ManiSDP.m) are four copies of one scheme: solve the manifold subproblem with
%   trustregions(), measure the KKT residues, cut the rank, add escape directions, adapt sigma.  Here the
%   scheme exists once, parameterised by a small trait record per kind, and every n x n quantity of the
%   reference's loop body (X = Y'*Y, A*x, eS, S, eig(S), svd(Y)) is replaced by a call into
%   libmanisdp_hip through manisdp_mex:
%       trustregions(problem, Y, opts)        -> 'set_point' + 'rtr' + 'get_point'   (whole RTR/tCG on the GPU)
%       c'*x, A*x                             -> 'al_primal'
%       eS, z, S                              -> 'al_dual'                             (S stays on the GPU)
%       eig(S): lambda_min, lambda_max, <= delta bottom eigenvectors -> 'escape_eigs' / 'escape_eigs_dual'
%       co() inside line_search               -> 'linesearch_cost' / 'linesearch_accept'
%       svd(Y) for the rank estimate          -> eig of the p x p Gram matrix (host, p is small); for onlyunitdiag the Gram
%                                                matrix itself, the rank cut and the widening by the escape directions
%                                                are 'factor_gram' / 'factor_rotate' / 'factor_append': the factor stays
%                                                on the GPU between two trustregions() calls
%   Option names, defaults, the printed protocol and the fields of DATA are the reference's (README.md:19-111 of
%   the reference; SURVEY.md appendix A).  Extra optional fields: options.Y0 (start point, any kind),
%   options.eig_tol / options.eig_maxit (Lanczos controls), options.dense_output (return dense X and S; default
%   n <= 8192 -- beyond that X is returned as the factor and data.S is left empty).
%
%   kind      'onlyunitdiag' | 'unitdiag' | 'unittrace' | 'generic'
%   prob      struct: .n, and either .C (onlyunitdiag) or .At, .b, .c (affine kinds)
%   defaults  N x 2 cell {name, value} of the entry point's option defaults

opt = options;
for q = 1:size(defaults, 1)
    if ~isfield(opt, defaults{q, 1}), opt.(defaults{q, 1}) = defaults{q, 2}; end
end
T = kind_traits(kind);
n = prob.n;
if T.affine, m = numel(prob.b); else, m = n; end

fprintf('ManiSDP is starting...\n');
fprintf('SDP size: n = %i, m = %i\n', n, m);

% ---- device handle (released when this function's workspace goes away, error or not)
if T.affine
    h = manisdp_mex(['create_' kind], prob.At, prob.b, prob.c, n);
    bvec = full(prob.b(:));
    scale_b = 1 + norm(bvec);
    y = zeros(m, 1);
    sigma = opt.sigma0;
else
    h = manisdp_mex('create_onlyunitdiag', prob.C);
end
release = onCleanup(@() manisdp_mex('destroy', h)); %#ok<NASGU>

% ---- start point (the reference lets trustregions draw it with M.rand())
p = opt.p0;
if isfield(opt, 'Y0') && ~isempty(opt.Y0)
    Y = opt.Y0;
elseif T.wide
    Y = T.normalise(randn(p, n));
else
    Y = T.normalise(randn(n, p));
end
U = [];
tr_opts = struct('maxiter', opt.TR_maxiter, 'maxinner', opt.TR_maxinner, 'tolgradnorm', opt.tolgradnorm);
if ~isfield(opt, 'eig_tol'),   opt.eig_tol = 1e-10; end
if ~isfield(opt, 'eig_maxit'), opt.eig_maxit = 20000; end
if ~isfield(opt, 'dense_output'), opt.dense_output = (n <= 8192); end

% onlyunitdiag without line search: the factor is re-shaped on the device and comes to the host once, at the end
keep_on_device = strcmp(kind, 'onlyunitdiag') && opt.line_search ~= 1;
resident = false;
data.status = 0;
if strcmp(kind, 'unitdiag'), fac_size = []; end
watch = [];                        % residues remembered for the slow-progress test
eta = inf;  certified = true;
gap = [];  pinf = [];  z = [];
t0 = tic;

for iter = 1:opt.AL_maxiter
    if strcmp(kind, 'unitdiag'), fac_size(end + 1) = p; end %#ok<AGROW>
    if T.affine, manisdp_mex('set_multipliers', h, y, sigma); end
    if ~resident, manisdp_mex('set_point', h, Y); end
    if ~isempty(U), backtrack_on_device(h, U); end
    info = manisdp_mex('rtr', h, tr_opts);
    gradnorm = info.gradnorm;
    if ~keep_on_device
        Y = manisdp_mex('get_point', h);
        Yeval = Y;                             % the point the residues below belong to; what X is built from at the end
    end

    % ---- KKT quantities from the device
    if T.affine
        [obj, Ax] = manisdp_mex('al_primal', h);
        resid = Ax - bvec;
        pinf = norm(resid)/scale_b;
        y = y - sigma*resid;
        z = manisdp_mex('al_dual', h, y);
        by = bvec'*y + sum(z);                 % sum([]) = 0 for the generic kind
        [lam, V, lam_top, okflag] = manisdp_mex('escape_eigs_dual', h, opt.delta, opt.eig_tol, opt.eig_maxit);
        gap = abs(obj - by)/(abs(by) + abs(obj) + 1);
    else
        z = manisdp_mex('get_z', h);
        obj = sum(z);
        [lam, V, lam_top, okflag] = manisdp_mex('escape_eigs', h, opt.delta, opt.eig_tol, opt.eig_maxit);
    end
    certified = (okflag ~= 0);                 % a Lanczos run that ran out of steps certifies nothing
    dinf = max(0, -lam(1))/(1 + lam_top);
    if T.affine, eta_now = max([gap, pinf, dinf]); else, eta_now = dinf; end
    if certified && (eta_now < opt.tol || iter == opt.AL_maxiter)
        % The regular escape call is warm-started (columns of Y and the previous call's vectors in its start block, or
        % span(Y) deflated): fast, but its lambda_min is an estimate.  Before dinf may end the solve -- always; the
        % estimate of the same call is never taken as a certificate -- and on the last pass, so that the reported dinf
        % is the true one, lambda_min is recomputed by a cold-started, undeflated run on S.
        [lam1, v1, top1, okflag] = independent_lambda_min(h, T.affine, n, opt);
        certified = (okflag ~= 0);
        dcheck = max(0, -lam1)/(1 + top1);
        if dcheck >= opt.tol && dinf < opt.tol
            V = [v1, V(:, 1:max(opt.delta - 1, 0))];  lam = [lam1; lam(1:end-1)];
        end
        dinf = dcheck;
    end

    % ---- numerical rank of the factor from its p x p Gram matrix
    if keep_on_device, G = manisdp_mex('factor_gram', h); elseif T.wide, G = Y*Y'; else, G = Y'*Y; end
    [Qg, wg] = eig((G + G')/2, 'vector');
    [wg, order] = sort(max(wg, 0), 'descend');
    Qg = Qg(:, order);
    sv = sqrt(wg);
    r = sum(sv >= opt.theta*sv(1));

    if T.affine
        fprintf('Iter %d, obj:%0.8f, gap:%0.1e, pinf:%0.1e, dinf:%0.1e, gradnorm:%0.1e, r:%d, p:%d, sigma:%0.3f, time:%0.2fs\n', ...
                iter, obj, gap, pinf, dinf, gradnorm, r, p, sigma, toc(t0));
        eta = max([gap, pinf, dinf]);
        now = [gap, pinf, dinf];
    else
        fprintf('Iter %d, obj:%0.8f, dinf:%0.1e, r:%d, p:%d, time:%0.2fs\n', iter, obj, dinf, r, p, toc(t0));
        eta = dinf;
        now = dinf;
    end
    if eta < opt.tol && certified
        fprintf('Optimality is reached!\n');
        if keep_on_device, Yeval = manisdp_mex('get_point', h); end
        break;
    end
    if mod(iter, T.watch_every) == 0
        if iter > T.watch_after && ~isempty(watch) && all(now > watch)
            data.status = 2;
            fprintf('Slow progress!\n');
            if keep_on_device, Yeval = manisdp_mex('get_point', h); end
            break;
        end
        watch = now;
    end

    % ---- rank cut, then escape directions (or the line-search direction for the next round)
    nneg = min(sum(lam < 0), opt.delta);       % missing pairs come back as +inf
    if T.at_least_one, nneg = max(nneg, 1); end
    if keep_on_device
        if iter == opt.AL_maxiter, Yeval = manisdp_mex('get_point', h); end     % last pass: the evaluated point
        if r <= p - 1
            manisdp_mex('factor_rotate', h, Qg(:, 1:r)');
            p = r;
        end
        try
            manisdp_mex('factor_append', h, V(:, 1:nneg), opt.alpha, 1);
            p = p + nneg;
            resident = true;
            continue;
        catch err
            % wider than the buffers the handle has allocated: finish this step on the host, re-enter through set_point
            if isempty(strfind(err.message, 'allocated capacity')), rethrow(err); end %#ok<STREMP>
            Y = manisdp_mex('get_point', h);   % already cut
            Y = T.normalise([Y; opt.alpha*V(:, 1:nneg)']);
            p = p + nneg;
            resident = false;
            continue;
        end
    end
    if r <= p - 1
        if T.wide, Y = Qg(:, 1:r)'*Y; else, Y = Y*Qg(:, 1:r); end
        p = r;
    end
    D = V(:, 1:nneg);                          % n x nneg
    if opt.line_search == 1
        if T.wide
            U = [zeros(p, n); D'];   Y = [Y; zeros(nneg, n)];
        else
            U = [zeros(n, p), D];    Y = [Y, zeros(n, nneg)];
        end
    else
        if T.wide, Y = [Y; opt.alpha*D']; else, Y = [Y, opt.alpha*D]; end
        Y = T.normalise(Y);
        U = [];
    end
    p = p + nneg;

    if T.affine
        if pinf < opt.tau1*gradnorm
            sigma = max(sigma/opt.gama, opt.sigma_min);
        elseif pinf > opt.tau2*gradnorm
            sigma = min(sigma*opt.gama, opt.sigma_max);
        end
    end
end

% ---- outputs in the reference's shape
if opt.dense_output
    if T.wide, X = Yeval'*Yeval; else, X = Yeval*Yeval'; end
    if T.affine
        data.S = manisdp_mex('get_dual_slack', h);
    else
        data.S = prob.C - spdiags(z(:), 0, n, n);
    end
else
    X = Yeval;  data.S = [];
end
data.X = X;
data.z = z;
if T.affine, data.y = y; data.gap = gap; data.pinf = pinf; end
data.dinf = dinf;
data.gradnorm = gradnorm;
data.time = toc(t0);
if strcmp(kind, 'unitdiag'), data.fac_size = fac_size; end
if data.status == 0 && (eta > opt.tol || ~certified)
    data.status = 1;
    fprintf('Iteration maximum is reached!\n');
end
fprintf('ManiSDP: optimum = %0.8f, time = %0.2fs\n', obj, toc(t0));
end

% -------------------------------------------------------------------------
function T = kind_traits(kind)
% What distinguishes the four entry points inside the common loop.
%   wide          factor is p x n with unit columns (oblique manifold); otherwise n x p
%   normalise     put a raw factor back on the manifold
%   at_least_one  always add one escape direction (ManiSDP_onlyunitdiag.m:74, ManiSDP_unitdiag.m:97)
%   watch_*       period and warm-up of the slow-progress test
T.affine = ~strcmp(kind, 'onlyunitdiag');
switch kind
    case {'onlyunitdiag', 'unitdiag'}
        T.wide = true;
        T.normalise = @(Z) Z./sqrt(sum(Z.^2, 1));
        T.at_least_one = true;
    case 'unittrace'
        T.wide = false;
        T.normalise = @(Z) Z/norm(Z, 'fro');
        T.at_least_one = false;
    case 'generic'
        T.wide = false;
        T.normalise = @(Z) Z;
        T.at_least_one = false;
    otherwise
        error('ManiSDP:hip:kind', 'unknown problem kind ''%s''', kind);
end
if strcmp(kind, 'unitdiag')
    T.watch_every = 50;  T.watch_after = 100;
else
    T.watch_every = 20;  T.watch_after = 50;
end
end

% -------------------------------------------------------------------------
function [lam1, v1, top1, okflag] = independent_lambda_min(h, affine, n, opt)
% Affine kinds with a dense S of moderate order: the reference's own eig(S) on the S the device holds.  Otherwise
% a cold-started run on the device: nothing deflated, nothing reused from earlier calls -- the block eigen-solver from
% hashed noise (sparse C), or plain Lanczos from a random combination of the columns of Y plus noise; twice if the first
% budget runs out.
if ~isfield(opt, 'verify_dense_max'), opt.verify_dense_max = 4000; end
if affine && n <= opt.verify_dense_max
    S = manisdp_mex('get_dual_slack', h);
    [W, w] = eig((S + S')/2, 'vector');
    lam1 = w(1);  v1 = W(:, 1);  top1 = w(end);  okflag = 1;
    return;
end
manisdp_mex('set_option', h, 'escape_deflate', 0);
manisdp_mex('set_option', h, 'escape_warm', 0);
% block eigen-solver (sparse C): hashed noise only; Lanczos path: span(Y) + 5 % noise
manisdp_mex('set_option', h, 'escape_start_y', double(manisdp_mex('escape_method', h) ~= 1));
restore = onCleanup(@() cellfun(@(nm, v) manisdp_mex('set_option', h, nm, v), {'escape_deflate', 'escape_warm', 'escape_start_y'}, {1, 1, 0})); %#ok<NASGU>
if affine, cmd = 'escape_eigs_dual'; else, cmd = 'escape_eigs'; end
[lam1, v1, top1, okflag] = manisdp_mex(cmd, h, 1, opt.eig_tol, opt.eig_maxit);
if ~okflag
    [lam1, v1, top1, okflag] = manisdp_mex(cmd, h, 1, opt.eig_tol, 4*opt.eig_maxit);
end
end

% -------------------------------------------------------------------------
function backtrack_on_device(h, U)
% The reference's line_search (step 1, factor 0.8, at most 15 reductions, sufficient decrease 1e-3) with every
% trial cost evaluated on the GPU; the accepted trial point becomes the resident point.
base = manisdp_mex('linesearch_cost', h, [], 0);
step = 1;
trial = manisdp_mex('linesearch_cost', h, U, step);
tries = 1;
while tries <= 15 && trial - base > -1e-3
    step = 0.8*step;
    trial = manisdp_mex('linesearch_cost', h, U, step);
    tries = tries + 1;
end
manisdp_mex('linesearch_accept', h);
end
